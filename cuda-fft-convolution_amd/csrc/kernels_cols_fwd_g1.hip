// kernels_cols_fwd_g1.hip -- forward-column kernels, configurations of group 1 of fast_paths.hpp
// (the kernel families are spread over translation units only to compile in parallel: make -j).
#define FC_TU_GROUP 1
#include "kernels_cols_fwd.inc"
