// kernels_cols_g0.hip -- output-column kernels, configurations of group 0 of fast_paths.hpp
// (the kernel families are spread over translation units only to compile in parallel: make -j).
#define FC_TU_GROUP 0
#include "kernels_cols.inc"
