// kernels.hpp -- host-callable launchers of the HIP kernels (kernels*.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "fast_paths.hpp"
#include "kernels_body.hpp"

namespace fc {

// Raises the dynamic-LDS limit of every kernel to the full 160 KiB of a gfx950 CU.
hipError_t kernels_init();

// dst plane = src plane reversed (flip of a column-major kh x kw plane along both axes)
hipError_t launch_flip_planes(const float* src, float* dst, int plane_elems, long nplanes, hipStream_t s);
// dst[map][x0 + x][y0 + y] += src[map][x][y], clipped to the dst_h x dst_w window (column-major maps)
hipError_t launch_add_window(float* dst, int dst_h, int dst_w, size_t dst_map_stride, int y0, int x0, const float* src, int src_h,
                             int src_w, size_t src_map_stride, int nmaps, hipStream_t s);
// dst[map] (dst_h x dst_w, contiguous) = window of src[map] at (off_h, off_w); src has src_h rows per column
hipError_t launch_crop_maps(const float* src, int src_h, size_t src_map_stride, float* dst, int dst_h, int dst_w, size_t dst_map_stride,
                            int off_h, int off_w, int nmaps, hipStream_t s);
// natural [f][fw][ch] <-> internal image-spectrum order (see kernels.hip: k_spectrum_reorder)
hipError_t launch_spectrum_reorder(bool to_natural, c32* S, size_t s_plane, int s_pitch, c32* nat, int fw, int ch, int planes,
                                   const int* row_of, const int* col_of, float scale, hipStream_t s);
// dst[map] (dst_h x dst_w >= src) = src[map] in the top-left corner, zero elsewhere
hipError_t launch_pad_maps(const float* src, int src_h, int src_w, size_t src_map_stride, float* dst, int dst_h, int dst_w,
                           size_t dst_map_stride, int nmaps, hipStream_t s);
hipError_t launch_cols_r2c(const ColsR2CArgs& a, int tiles, int planes, int threads, size_t lds_bytes, hipStream_t s);
hipError_t launch_rows_fwd(const RowsFwdArgs& a, int rows, int threads, size_t lds_bytes, hipStream_t s);
// specialised forward image rows (fast_rows_fwd.hpp); hipErrorInvalidValue if L has no configuration
hipError_t launch_fast_rows_fwd(int L, const FastRowsFwdArgs& a, int rows, hipStream_t s);
hipError_t launch_spectral_rows(const SpectralRowsArgs& a, int rows, int kernels, int threads, size_t lds_bytes, hipStream_t s);
// fast path (fast_rows.hpp); hipErrorInvalidValue if no instantiation matches (L, nz2)
hipError_t launch_fast_rows(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, hipStream_t s);
// several maps per workgroup (fast_rows_multi.hpp, F = 1): kernels_per_wg consecutive kernels share one fetch of the image-spectrum row
hipError_t launch_fast_rows_multi(int L, int nz2, const FastRowsArgs& a, int rows, int kernels, int kernels_per_wg, hipStream_t s);
// how many workgroups of that kernel a CU holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor)
hipError_t fast_rows_multi_wgs_per_cu(int L, int nz2, const FastRowsArgs& a, int* wgs_per_cu);
hipError_t launch_fast_cols(int M, int T, const FastColsArgs& a, int num_cus, hipStream_t s);
hipError_t launch_fast_cols_fwd(int M, int T, bool pruned, const FastColsFwdArgs& a, int num_cus, hipStream_t s);
// image columns (full variant) and kernel columns (pruned or full) of one plan in ONE launch (kernels_cols_fwd.hip)
hipError_t launch_fast_cols_fwd_pair(int M, int T, const FastColsFwdArgs& image, const FastColsFwdArgs& kernels, bool kernels_pruned,
                                     int num_cus, hipStream_t s);
hipError_t launch_cols_c2r(const ColsC2RArgs& a, int tiles, int kernels, int threads, size_t lds_bytes, hipStream_t s);

}  // namespace fc
