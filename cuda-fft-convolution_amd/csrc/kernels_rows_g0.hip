// kernels_rows_g0.hip -- one-map spectral-row and forward image-row kernels, configurations of group 0 of fast_paths.hpp
// (the kernel families are spread over translation units only to compile in parallel: make -j).
#define FC_TU_GROUP 0
#include "kernels_rows.inc"
