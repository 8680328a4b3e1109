// kernels_cols_g1.hip -- output-column kernels, configurations of group 1 of fast_paths.hpp
// (the kernel families are spread over translation units only to compile in parallel: make -j).
#define FC_TU_GROUP 1
#include "kernels_cols.inc"
