// kernels_rows_multi_g1.hip -- multi-map spectral-row kernels, configurations of group 1 of fast_paths.hpp
// (the kernel families are spread over translation units only to compile in parallel: make -j).
#define FC_TU_GROUP 1
#include "kernels_rows_multi.inc"
