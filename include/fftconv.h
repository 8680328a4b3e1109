/*
 * fftconv.h -- C ABI of the MI355X-native 2-D FFT-convolution engine (libfftconv.so).
 *
 * This is the drop-in boundary for the reference's `cudaConvolutionFFT` hot path
 * (chrischoy/CUDA-FFT-Convolution).  Every entry point names the reference interface it
 * replaces (paths relative to the reference tree).  Plain C: pointers and sizes only, no
 * torch / HIP types in the signatures (a HIP stream is passed as an opaque void*).
 *
 * Conventions shared by all entry points
 *   - Arrays use MATLAB column-major layout exactly as the reference reads them:
 *     data is H x W x F, element (y, x, z) at z*H*W + x*H + y
 *     (src/cudaConvFFTData.cuh:26-27); kernel k is kh x kw x F in the same layout.
 *   - Every result map is the FULL padded window FFT_H x FFT_W (column-major, NOT cropped),
 *     FFT_X = fftconv_fft_size16(DATA_X + MAX_KERNEL_X - 1)
 *     (src/cudaConvolutionFFT.cu:103-110,198-200).  Cells outside the linear-convolution
 *     support (DATA+k-1) are zero up to fp32 round-off, as in the reference.
 *   - The product is a plain complex multiply (convolution, not correlation:
 *     src/cudaConvFFTData.cuh:62-65); flip the kernel for template matching
 *     (demoCudaConvolutionFFT.m:63-69).
 *   - Functions return FFTCONV_OK (0) or a negative fftconv_status; the message of the last
 *     failure on the calling thread is returned by fftconv_last_error().  The library never
 *     calls exit() (the reference's CUDA_SAFE_CALL does: src/cudaConvFFTData.h:6-29) and
 *     releases all device memory on error paths.
 *   - There is no CPU fallback: without a usable HIP device every compute entry point fails
 *     with FFTCONV_ERR_NO_DEVICE / FFTCONV_ERR_HIP.
 */
#ifndef FFTCONV_H
#define FFTCONV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fftconv_status {
    FFTCONV_OK = 0,
    FFTCONV_ERR_INVALID_ARG = -1,      /* "Wrong number of inputs" / "Invalid data input" class (src/cudaConvolutionFFT.cu:45-54) */
    FFTCONV_ERR_THREAD_SIZE = -2,      /* thread-size array does not have 4 elements (src/cudaConvolutionFFT.cu:72-73) */
    FFTCONV_ERR_KERNEL_SHAPE = -3,     /* feature mismatch or kernel larger than the FFT window (src/cudaConvolutionFFT.cu:242-243) */
    FFTCONV_ERR_KERNEL_EXCEEDS_MAX = -4, /* kernel larger than MAX_KERNEL and the internal transform is not the ceil16 window */
    FFTCONV_ERR_UNSUPPORTED_SIZE = -5, /* transform does not fit the single-pass LDS engine */
    FFTCONV_ERR_NO_DEVICE = -6,
    FFTCONV_ERR_HIP = -7,              /* a HIP runtime call failed; message has the HIP error string */
    FFTCONV_ERR_ALLOC = -8,
    FFTCONV_ERR_NO_IMAGE = -9          /* convolve called before an image spectrum exists */
} fftconv_status;

/* Error id the MEX gateway raises for argument errors (src/cudaConvolutionFFT.cu:30). */
#define FFTCONV_MEX_ERROR_ID "cudaConvFFTData:InvalidInput"

enum { FFTCONV_HOST = 0, FFTCONV_DEVICE = 1,
       FFTCONV_AUTO = 2 /* kernels of fftconv_plan_convolve only: every pointer may be host or device memory, the
                           HIP runtime tells them apart (a cell mixing host arrays and gpuArrays:
                           src/cudaConvolutionFFT.cu:207-238) */ };

/* computeFFTsize16 (src/cudaConvFFTData.h:96-102): round up to a multiple of 16. */
int fftconv_fft_size16(int data_size);

/* computeFFTsize (src/cudaConvFFTData.h:67-94), the reference's alternative sizing: up to a
 * multiple of 16, then up to the next power of two.  Its shipped path never calls it; plans offer
 * it as "output_region" 4. */
int fftconv_fft_size_pow2(int data_size);

/* Message of the last failure on this thread ("" if none). */
const char *fftconv_last_error(void);

/* Library version string. */
const char *fftconv_version(void);

/* Number of visible HIP devices (0 and FFTCONV_ERR_NO_DEVICE when there is none). */
int fftconv_device_count(int *count);

/* ------------------------------------------------------------------------------------------
 * One-shot entry: the body of mexFunction in src/cudaConvolutionFFT.cu:27-311.
 *
 *   cvcell = cudaConvolutionFFT(data, maxKernelH, maxKernelW, kernelCell[, threadSize][, gpuId])
 *
 *   data, data_h, data_w, feature_dim   prhs[0]  host single H x W x F (:49-54,92-99); F = 1 is
 *                                                accepted (the reference rejects it)
 *   max_kernel_h, max_kernel_w          prhs[1], prhs[2] (:58-59)
 *   n_kernel, kernels, kernel_h/_w      prhs[3]  cell array of host single kh x kw x F (:64-67,207-222)
 *   kernel_f                            per-kernel feature count for the :242 check; NULL = F
 *   thread_size, n_thread_size          prhs[4]  optional; must have 4 elements (:72-73); the
 *                                                values are accepted and ignored (block shapes
 *                                                are the engine's business)
 *   gpu_id                              prhs[5]  0-based device (:85-89); < 0 = current device
 *   out                                 plhs[0]  n_kernel caller buffers of FFT_H*FFT_W floats
 *                                                (the reference allocates them: :284-288)
 *   fft_h, fft_w                        out, nullable: the window size
 * Synchronous: results are complete on return.  The plan (tables, device scratch) goes back into the
 * plan cache below instead of being torn down; fftconv_cache_clear() releases it.
 * Images whose padded size exceeds what one plan transforms in a single pass, and large images that
 * run faster that way, are convolved block-wise (fftconv_plan_options.blockwise): blocks of a
 * shorter transform, each block's rectangle of the FFT_H x FFT_W maps stored by the output kernel
 * (kernels larger than MAX_KERNEL are rejected on that path).
 * ------------------------------------------------------------------------------------------ */
int fftconv_convolution_fft(const float *data, int data_h, int data_w, int feature_dim,
                            int max_kernel_h, int max_kernel_w,
                            int n_kernel, const float *const *kernels,
                            const int *kernel_h, const int *kernel_w, const int *kernel_f,
                            const double *thread_size, int n_thread_size,
                            int gpu_id,
                            float *const *out, int *fft_h, int *fft_w);
/* The same with plan options (forward-declared below; NULL = defaults) and kernels that may be
 * device-resident: kernel_location FFTCONV_HOST, FFTCONV_DEVICE, or FFTCONV_AUTO for a cell that
 * mixes host arrays and gpuArrays as the reference's loop allows (src/cudaConvolutionFFT.cu:207-238). */
struct fftconv_plan_options;
int fftconv_convolution_fft_ex(const float *data, int data_h, int data_w, int feature_dim,
                               int max_kernel_h, int max_kernel_w,
                               int n_kernel, const float *const *kernels,
                               const int *kernel_h, const int *kernel_w, const int *kernel_f,
                               int kernel_location,
                               const double *thread_size, int n_thread_size,
                               int gpu_id,
                               float *const *out, int *fft_h, int *fft_w,
                               const struct fftconv_plan_options *options);

/* ------------------------------------------------------------------------------------------
 * Plan cache of the one-shot entries.  The reference builds its cuFFT plans, allocates its six device
 * buffers and tears everything down inside every MEX call (src/cudaConvolutionFFT.cu:127-142,144-185,
 * 302-310); here that state is a plan, and fftconv_convolution_fft / _ex (and with them the
 * cudaConvolutionFFT gateway) keep the plans of their last few distinct problems -- key: data size, F,
 * MAX_KERNEL, device, creation options -- with their tables, device scratch and host copy threads, so that
 * a repeated call pays for the transfers and the kernels only (cfg2: 15.9 ms -> the reused-plan time).
 * A plan is handed to one call at a time; a concurrent call with the same key builds a plan of its own.
 * A call that fails with a HIP / allocation error does not return its plan to the cache.
 *   fftconv_cache_configure  max_plans: plans kept (default 4; 0 switches the cache off and empties it),
 *                            max_bytes: device memory the idle cached plans may hold together (default: a quarter of
 *                            the device's memory, at most 48 GiB; 0 = keep the current value); least recently used plans go
 *                            first.  Whatever the limits, EVERY idle plan is released -- and the request repeated once -- when
 *                            a device allocation of this library fails (any entry, not only the one-shot ones): cached
 *                            scratch never turns a call that used to fit into an out-of-memory error
 *   fftconv_cache_clear      destroys every idle cached plan (device memory, host threads).  The gateways
 *                            register it with mexAtExit; a process that unloads the library calls it first.
 *   fftconv_cache_stats      nullable outputs: plans held now, hits / misses so far, device bytes held
 * ------------------------------------------------------------------------------------------ */
int fftconv_cache_configure(int max_plans, size_t max_bytes);
int fftconv_cache_clear(void);
int fftconv_cache_stats(long *plans, long *hits, long *misses, size_t *device_bytes);

/* Where the time of this thread's last one-shot call (fftconv_convolution_fft / _ex) went, milliseconds of wall
 * clock: what the reference spends in plan creation and allocation (src/cudaConvolutionFFT.cu:127-142,144-185), the
 * image upload + transform (:146-169), the per-kernel loop with its copies (:204-291) and the teardown (:302-310).
 * With plan option / fftconv_plan_options.verbose the same line goes to stderr. */
typedef struct fftconv_call_timing {
    double plan_ms;        /* cache lookup, or plan creation: tables, kernel set-up, device tables */
    double image_ms;       /* the image staged / copied and its transform QUEUED (small images go through pinned staging and the
                              call does not wait for the GPU: the transform's run time then shows in convolve_ms) */
    double convolve_ms;    /* kernel uploads, every map computed and copied into the caller's buffers */
    double release_ms;     /* plan back into the cache, or its teardown */
    double total_ms;
    int cache_hit;         /* 1: the plan came from the cache */
} fftconv_call_timing;
int fftconv_last_call_timing(fftconv_call_timing *timing);

/* ------------------------------------------------------------------------------------------
 * Plan API: the state the reference keeps inside one mexFunction call (cuFFT plans
 * src/cudaConvolutionFFT.cu:122-142, image spectrum d_CFFT_DATA :165-168, scratch :182-185)
 * and, in its multi-GPU sketch, inside a ConvPlan (src/cudaConvFFTDataStreams.cu:273-328),
 * made a first-class object so the image spectrum is computed once and reused against any
 * number of kernels and calls (what cudaFFTData / cudaConvFFTData exist for:
 * src/cudaFFTData.cu:18-160, src/cudaConvFFTData.cu:24-306).
 * A plan is bound to one device; calls on one plan must not overlap.
 * ------------------------------------------------------------------------------------------ */
typedef struct fftconv_plan fftconv_plan;

typedef struct fftconv_plan_info {
    int data_h, data_w, feature_dim;
    int max_kernel_h, max_kernel_w;
    int fft_h, fft_w;          /* output window (reference's ceil16 sizes) */
    int transform_h, transform_w; /* internal transform lengths (>= DATA + MAXK - 1) */
    int spectrum_rows;         /* transform_h / 2 + 1 */
    int spectrum_pitch;        /* complex elements per spectrum row */
    int gpu_id;
    int exact_window;          /* 1 if transform == window: circular modulus equals the reference's */
    size_t spectrum_bytes;     /* size of the image spectrum buffer */
    size_t map_bytes;          /* fft_h * fft_w * sizeof(float) */
    size_t workspace_bytes;    /* device scratch currently held */
    int out_h, out_w;          /* size of every result map: the window, or the "output_region" chosen */
    size_t out_map_bytes;      /* out_h * out_w * sizeof(float) */
} fftconv_plan_info;

/* hip_stream: hipStream_t to run on (NULL = the device's default stream). */
int fftconv_plan_create(fftconv_plan **plan, int data_h, int data_w, int feature_dim,
                        int max_kernel_h, int max_kernel_w, int gpu_id, void *hip_stream);

/* Choices fixed at plan creation.  Zero-initialise, set struct_size = sizeof(fftconv_plan_options),
 * then the fields wanted; NULL options = all defaults.  The library reads no environment
 * variables: everything that steers it goes through this struct or fftconv_plan_set_option. */
typedef struct fftconv_plan_options {
    size_t struct_size;
    int kernel_path;    /* 0 (default): specialised kernels where the transform lengths have them, tiled
                         *    intermediate; 1: generic kernels only (any supported length; the cross-check of
                         *    the specialised path); 2: specialised kernels with a row-major intermediate */
    int rows_group;     /* maps one workgroup of the spectral-row kernel walks with its image-spectrum row in
                         *    registers (F = 1): 0 (default) chosen per launch, 1 one map per workgroup, n > 1 fixed */
    int max_transform;  /* > 0: largest transform length a plan may use; beyond it the plan is block-wise
                         *    with block transforms of at most this size (see `blockwise`) */
    int exact_window;   /* 1: the transform lengths must be the ceil16 window FFT_H x FFT_W itself (the
                         *    reference's circular modulus, plan_info.exact_window = 1) -- what
                         *    fftconv_plan_export_spectrum / _import_spectrum need; creation fails with
                         *    FFTCONV_ERR_UNSUPPORTED_SIZE when the window has a prime factor above 17 */
    int blockwise;      /* 0 (default): the plan convolves block-wise where that is needed or faster -- padded sizes beyond one
                         *    LDS-resident transform pass (about 20 000 samples along w) or beyond max_transform, and sizes whose
                         *    one-pass transform would run on the slower long-transform kernels (from about 4900 x 4900 with
                         *    63 x 63 kernels; the planner weighs measured costs per length).  Blocks are overlap-save over a
                         *    block plan whose transform has specialised kernels: every block is a piece of the image with
                         *    MAX_KERNEL - 1 history rows / columns, convolved circularly, and the output kernel stores the part
                         *    that is not wrapped straight into the block's rectangle of the maps (nothing is summed, every
                         *    element is written once); where the block transform has no specialised kernels (kernel_path = 1,
                         *    very wide kernels) blocks are zero-padded and summed (overlap-add).  Block spectra are kept per
                         *    image; every plan entry point works (two-step, packed device-resident, multi-device), except the
                         *    spectrum exchange in the reference's order and "output_region"; kernels larger than MAX_KERNEL
                         *    are rejected there.  1: never block-wise -- one pass, or FFTCONV_ERR_UNSUPPORTED_SIZE.  (Appended
                         *    in 0.2: a struct_size without this field is accepted and means 0.)  fftconv_plan_get_option
                         *    "blockwise" reads the number of blocks of a plan (0 = single pass), "overlap_save" whether they are
                         *    stored (1) or summed (0); plan_info.transform_h / _w are the block transform then.  The reference
                         *    plans cuFFT for any size: src/cudaFFTData.cu:72-103, src/cudaConvFFTData.cu:92-98. */
    int verbose;        /* 1: the plan starts with option "verbose" on (the reference's compile-time `debug`,
                         *    src/cudaConvolutionFFT.cu:9), and a one-shot call prints where its time went
                         *    (fftconv_call_timing).  (Appended in 0.3; a struct_size without it means 0.) */
} fftconv_plan_options;
int fftconv_plan_create_ex(fftconv_plan **plan, int data_h, int data_w, int feature_dim,
                           int max_kernel_h, int max_kernel_w, int gpu_id, void *hip_stream,
                           const fftconv_plan_options *options);
/* 1 if `plan` is a live plan of this library (created and not yet destroyed), else 0.  The MEX
 * gateways check the uint64 handles MATLAB hands back with it before dereferencing them. */
int fftconv_plan_is_live(const fftconv_plan *plan);
int fftconv_plan_destroy(fftconv_plan *plan);
int fftconv_plan_get_info(const fftconv_plan *plan, fftconv_plan_info *info);

/* Zero-pad + forward transform of the image: padData + cufftExecR2C on the data
 * (src/cudaConvolutionFFT.cu:144-169; src/cudaFFTData.cu:105-147).
 * location: FFTCONV_HOST (pageable or pinned host memory) or FFTCONV_DEVICE.
 * Ordering: the transform is queued on the plan's stream and the call may return before it has run, for device AND host
 * input.  Host input: `data` has been consumed when the call returns (copied, or staged in the plan's pinned buffer -- images
 * up to 1 MiB -- in which case the call does not wait for the GPU at all); the spectrum is ordered on the plan's stream like
 * any other work of the plan.  A caller that reads fftconv_plan_spectrum() on ANOTHER stream orders that stream behind the
 * plan's (an event, or fftconv_plan_synchronize) whatever the location.  Device input: `data` must stay valid until the
 * stream has passed the transform. */
int fftconv_plan_set_image(fftconv_plan *plan, const float *data, int location);

/* Device pointer + size of the image spectrum (library-owned, valid until destroy).  This is the
 * buffer the multi-GPU path broadcasts (the reference's cudaMemcpyPeerAsync of d_CFFT_DATA,
 * src/cudaConvFFTDataStreams.cu:279-289): rank 0 fills it with set_image, the others receive it
 * and call fftconv_plan_mark_spectrum_valid.  The layout is internal. */
int fftconv_plan_spectrum(fftconv_plan *plan, void **device_ptr, size_t *bytes);
int fftconv_plan_mark_spectrum_valid(fftconv_plan *plan);
/* Make the plan keep its image spectrum in caller-owned device memory (>= spectrum_bytes, 16-byte
 * aligned, must outlive the plan or the next call of this function; NULL = back to the plan's own
 * buffer).  Lets a communication library that owns its buffers (torch.distributed / RCCL)
 * broadcast the spectrum with no staging copy.  Invalidates the current spectrum. */
int fftconv_plan_use_spectrum_buffer(fftconv_plan *plan, void *device_ptr, size_t bytes);

/* The image spectrum in the reference's own order: what cudaFFTData returns as a complex single
 * gpuArray of (FFT_H/2+1) x FFT_W x F (src/cudaFFTData.cu:90-103,150), i.e. cufftExecR2C's output
 * [f][FFT_W][FFT_H/2+1] (src/cudaConvolutionFFT.cu:122-142: the halved dimension is h), unnormalised,
 * interleaved (re, im) floats; spectrum_natural_bytes = 8 * F * FFT_W * (FFT_H/2+1).
 * export: after fftconv_plan_set_image; import: replaces set_image (a spectrum the caller kept or
 * edited).  Both need a plan whose transform is the window (fftconv_plan_options.exact_window, or
 * plan_info.exact_window = 1 anyway) and return FFTCONV_ERR_UNSUPPORTED_SIZE otherwise.
 * location: FFTCONV_HOST or FFTCONV_DEVICE; synchronous for host memory. */
int fftconv_plan_export_spectrum(fftconv_plan *plan, float *spectrum, int location);
int fftconv_plan_import_spectrum(fftconv_plan *plan, const float *spectrum, int location);

/* Per-kernel loop of src/cudaConvolutionFFT.cu:204-291 for arbitrary (possibly different)
 * kernel sizes.  kernels[k]: kh[k] x kw[k] x F, host or device (gpuArray kernels: :224-238);
 * out[k]: FFT_H*FFT_W floats, host or device.  Synchronous for host output. */
int fftconv_plan_convolve(fftconv_plan *plan, int n_kernel,
                          const float *const *kernels, const int *kernel_h, const int *kernel_w,
                          int kernel_location,
                          float *const *out, int out_location);

/* Same for n_kernel equally-sized kernels packed contiguously in device memory
 * ([n][F][kw][kh], i.e. n consecutive MATLAB arrays) writing n consecutive maps to device
 * memory ([n][FFT_W][FFT_H]).  Fully asynchronous on the plan's stream: this is the
 * device-resident mode the benchmark times. */
int fftconv_plan_convolve_packed(fftconv_plan *plan, int n_kernel, const float *kernels_device,
                                 int kernel_h, int kernel_w, float *out_device);

/* Optional split of fftconv_plan_convolve_packed: queues only the part that does not depend on
 * the image (the h-transform of the kernels' columns, i.e. padData + the H half of cufftExecR2C on
 * the kernels, src/cudaConvolutionFFT.cu:245-255) on the plan's stream, at once, so that it overlaps
 * the arrival of the image spectrum (the RCCL broadcast on the other ranks).  The next
 * fftconv_plan_convolve_packed call with the same arguments reuses it; any other use of the plan discards it.
 * With plan option "defer_prepare" = 1 the call only RECORDS the request: the pass then rides in the launch of
 * the next fftconv_plan_set_image on the same stream (one launch fewer per step: what small problems want
 * when the image transform follows on the same stream), or is launched by the next convolve / set_stream /
 * synchronize -- the caller gives up the overlap above for it. */
int fftconv_plan_prepare_kernels_packed(fftconv_plan *plan, int n_kernel, const float *kernels_device,
                                        int kernel_h, int kernel_w);

/* Re-bind the plan to another stream (NULL = the default stream).  Everything queued so far stays
 * on the old stream; the caller orders the two.  After one warm-up call with the same arguments
 * (which sizes the scratch buffers) fftconv_plan_set_image(DEVICE) and
 * fftconv_plan_convolve_packed allocate nothing and never synchronise, so they can be recorded
 * into a HIP graph: bind the plan to the capturing stream, capture, replay the graph
 * (bench.py --graph; the launch-bound small configurations gain most). */
int fftconv_plan_set_stream(fftconv_plan *plan, void *hip_stream);

/* Block until everything queued on the plan's stream has finished. */
int fftconv_plan_synchronize(fftconv_plan *plan);

/* Options: "batch_maps" (kernels per spectral/output launch, 0 = auto),
 *          "kernel_chunk_mb" (0 = auto: the kernels' column spectra are produced one launch's worth at a
 *             time, right before the row kernel reads them; > 0: as many launches' worth as fit that many MiB),
 *          "tune_placement" (k > 1: the next time the intermediate buffer is (re)allocated, k candidate
 *             allocations of it are timed with the output kernel writing into the caller's map buffer and
 *             the fastest is kept -- on MI355X the output kernel runs 3-4 % faster or slower depending on which
 *             physical allocations hold its two buffers; blocking, ~70-100 ms, once per
 *             allocation; transient device memory while it runs: k intermediates plus k - 1 spacer allocations of at
 *             most 12 GiB (an eighth of what is free) each; skipped inside a stream capture and on plans with an output
 *             window (the blocks of a block-wise plan); 0: never; -1 (default): automatic -- five candidates where a launch
 *             writes at least 2 GiB of maps (BASELINE cfg3 / cfg4) AND at least 60 % of the device's memory is free at that
 *             moment, nothing otherwise (and never in a one-shot call whose plan is not kept: cache switched off).  It pays for itself after ~600
 *             cfg3 steps (7 s of work on the plan).  Which state an untuned plan lands in is a property of the machine's allocation order:
 *             profiles/r05q_default_vs_tuned_final_library*.txt),
 *          "rows_group" (fftconv_plan_options.rows_group, changeable between calls),
 *          "dynamic_tiles" (1, the default for transforms of 864 points and more along h (M >= 432): the persistent workgroups of
 *             the output kernel take their tiles from a queue in device memory -- one counter per XCD, chunks of adjacent tiles,
 *             work stealing at the end -- instead of a fixed 1 / grid share each.  A workgroup whose CU is held by another
 *             kernel (the RCCL broadcast of a multi-GPU step, src/cudaConvFFTDataStreams.cu:279-289,338-447; another library)
 *             then delays nobody's tiles: beside 16-32 foreign workgroups a cfg4 step loses 2.5 % instead of 4.5-7 %, alone it
 *             runs equal or up to 3 % faster (profiles/r05a_contention_ab.txt).  0: the static deal -- the default of smaller
 *             transforms, whose tiles are too short for a ticket fetched one tile ahead (profiles/r05k_dynamic_tiles_by_size.txt);
 *             2: the forward column kernels take their tiles from a counter too (measured slower alone: A/B and tests).
 *             Results are identical bit for bit),
 *          "profile" (1: time every kernel launch with HIP events on the plan's stream),
 *          "profile_kinds" (bit mask over the indices of fftconv_profile: only those kinds are timed
 *             while "profile" is on, 0 = all -- lets a caller time the hot kernels inside its own timed
 *             region at the cost of two event records per launch),
 *          "host_stream" (how maps reach HOST output buffers -- the reference's blocking
 *             cudaMemcpy per map, src/cudaConvolutionFFT.cu:284-286: 0 = blocking copies after each
 *             batch; 1 (default) = host threads of the plan copy batch b straight into the caller's
 *             buffers while batch b+1 is computed; 2 = through a ring of pinned chunks),
 *          "host_threads", "host_chunk_kb", "host_slots" (shape of that machinery, 0 = auto),
 *          "host_min_kb" (maps smaller than this many KiB leave by blocking copies whatever "host_stream"
 *             says; default 1024: the copy threads buy small maps nothing),
 *          "host_pinned" (1, default: small HOST arrays go through pinned staging buffers of the plan -- an image of at most
 *             1 MiB and a kernel group of at most 512 KiB are copied by the CPU and read by the GPU from there (at most
 *             512 KiB: in place, no copy command), maps of at most 4 MiB that leave by blocking copies come back several
 *             per copy and are handed out by the CPU (a single map above 64 KiB keeps its plain copy); the reference issues
 *             one blocking cudaMemcpy per array, src/cudaConvolutionFFT.cu:148,231,286.  Cached one-shot calls of the
 *             reference's demo shape: 120 -> 75 us, profiles/r04s_small_call_latency.txt.  0: plain copies from / into
 *             the caller's arrays),
 *          "output_region" (which part of the padded window a result map holds, for the plan's
 *             MAX_KERNEL sizes K: 0 (default) the whole FFT_H x FFT_W window as the reference
 *             returns it; 1 "full" = the linear convolution, (DATA + K - 1), what the demo crops by
 *             hand (demoCudaConvolutionFFT.m:149); 2 "same" = DATA-sized, centred; 3 "valid" =
 *             DATA - K + 1, no zero padding involved; 4 "pow2 window" = fftconv_fft_size_pow2(DATA + K - 1)
 *             per dimension, the window the reference's unused computeFFTsize would give
 *             (src/cudaConvFFTData.h:67-94): the convolution in its top-left corner, zeros elsewhere.
 *             Result buffers then hold out_h x out_w floats (fftconv_plan_get_info)),
 *          "defer_prepare" (1: fftconv_plan_prepare_kernels_packed records its request instead of launching it, see
             there; 0 (default): launched at once.  Read-only "prepare_pending": 1 while such a request waits),
          "verbose" (1: the sizes and launch shapes of every stage go to stderr as the work is queued -- the
 *             reference's compile-time `debug` switch, src/cudaConvolutionFFT.cu:9,60,100,114,240,258; 0 (default) silent),
 *          "flip_kernels" (1: every kernel is flipped along h and w on the device before it is
 *             transformed, i.e. the plan correlates -- the "Flip Kernel (Required)" step of
 *             demoCudaConvolutionFFT.m:63-69 done here instead of in MATLAB; the reference keeps a
 *             conjugate-product variant commented out, src/cudaConvFFTData.cuh:42-45,63). */
int fftconv_plan_set_option(fftconv_plan *plan, const char *name, long value);
/* Current value of an option, plus read-only ones: "tuned_candidates" (allocations tried by the last placement tuning)
 * and "tuned_best" (index of the one kept), "blockwise" (blocks of a block-wise plan, 0 = one pass), "overlap_save" (1: the blocks are stored by the output kernel),
 * "rows_slots_per_cu" (workgroups of the multi-map row kernel a CU holds at once: what the walk length is chosen for),
 * "specialised_kernels" (bit 0: the spectral-row pass runs on a specialised kernel, bit 1: the column passes do; 3 = no
 * generic kernel runs for this plan). */
int fftconv_plan_get_option(fftconv_plan *plan, const char *name, long *value);

typedef struct fftconv_profile {
    /* accumulated since the last reset; index 0 kernel_cols (h forward of the kernels),
     * 1 spectral_rows, 2 cols_c2r (output), 3 image_cols, 4 image_rows */
    double ms[5];
    long launches[5];
    long units[5];  /* kernels (maps) processed by those launches */
} fftconv_profile;
int fftconv_plan_get_profile(fftconv_plan *plan, fftconv_profile *prof, int reset);

/* ------------------------------------------------------------------------------------------
 * Two-step API of the reference, as thin aliases of the plan API:
 *   fftData = cudaFFTData(data, kernelH, kernelW)        src/cudaFFTData.cu:18-160
 *   cvcell  = cudaConvFFTData(fftData, kernelCell[, threadSize])  src/cudaConvFFTData.cu:24-306
 * The handle plays the role of the complex gpuArray the reference returns.
 * ------------------------------------------------------------------------------------------ */
int fftconv_fft_data(const float *data, int data_h, int data_w, int feature_dim,
                     int kernel_h, int kernel_w, int gpu_id, fftconv_plan **fft_data);
/* (kernels of fftconv_conv_fft_data: host arrays or device-resident ones, told apart by the HIP
 *  runtime -- FFTCONV_AUTO -- as src/cudaConvFFTData.cu accepts gpuArray cells too) */
int fftconv_conv_fft_data(fftconv_plan *fft_data, int n_kernel, const float *const *kernels,
                          const int *kernel_h, const int *kernel_w, const int *kernel_f,
                          const double *thread_size, int n_thread_size, float *const *out);

/* ------------------------------------------------------------------------------------------
 * Several GPUs from one process: the reference's multi-GPU / multi-stream sketch
 * (src/cudaConvFFTDataStreams.cu -- one ConvPlan per (GPU, stream) with private scratch :273-328,
 * the image spectrum copied from GPU 0 to GPU g with cudaMemcpyPeerAsync :279-289, kernels dealt
 * over the plans :338-447, a synchronisation barrier at the end :452-468; it never built or ran).
 * Here: one plan and one stream per listed device, the image transformed once on devices[0], its
 * spectrum copied to every other device over xGMI (hipMemcpyPeerAsync, each copy on the
 * destination's stream so that the links work in parallel), the kernels dealt in CONTIGUOUS
 * blocks (device g of n takes kernels [g*N/n, (g+1)*N/n), sizes differing by at most one), one
 * host thread per device while a call runs, everything complete on return.  The same device may
 * be listed more than once (two plans on one GPU: the reference's N_BATCH_PER_GPU = 2).
 * With one process per GPU (torch.distributed / RCCL) use the plan API and broadcast the
 * spectrum buffer instead: INTEGRATION.md.
 * ------------------------------------------------------------------------------------------ */
typedef struct fftconv_multi fftconv_multi;
int fftconv_multi_create(fftconv_multi **multi, int data_h, int data_w, int feature_dim,
                         int max_kernel_h, int max_kernel_w, const int *devices, int n_devices,
                         const fftconv_plan_options *options);
int fftconv_multi_destroy(fftconv_multi *multi);
/* After a successful fftconv_multi_create, fftconv_last_error() is "" -- or a warning naming the devices that have
 * no direct (xGMI peer) access to devices[0]: copies to those are staged by the runtime (slower, still correct).
 * Options: "spectrum_transport" 0 (default) = the reference's form, one peer copy per destination
 *             (src/cudaConvFFTDataStreams.cu:282-287); 1 = ONE collective: ncclBroadcast of the spectrum buffer over a
 *             communicator of the listed devices (RCCL over xGMI; librccl.so is loaded on first use, the devices must
 *             be distinct).  Where RCCL cannot be set up the copies are used and fftconv_last_error() says why;
 *          "verbose" 1 = what happens, to stderr (also sets every plan's "verbose").
 * Read-only: "transport_used" (what the last set_image / import_spectrum used), "peer_direct" (destinations with
 * direct access to devices[0]). */
int fftconv_multi_set_option(fftconv_multi *multi, const char *name, long value);
int fftconv_multi_get_option(const fftconv_multi *multi, const char *name, long *value);
/* padData + cufftExecR2C of the image on devices[0] (src/cudaConvolutionFFT.cu:144-169), then the
 * peer copies.  location FFTCONV_HOST, or FFTCONV_DEVICE for memory of devices[0]. */
int fftconv_multi_set_image(fftconv_multi *multi, const float *data, int location);
/* Instead of an image: its spectrum in the reference's order (fftconv_plan_import_spectrum; the handle
 * must have been created with fftconv_plan_options.exact_window = 1) -- what the reference's
 * multi-GPU MEX takes as its first argument (src/cudaConvFFTDataStreams.cu:160-187). */
int fftconv_multi_import_spectrum(fftconv_multi *multi, const float *spectrum, int location);
/* The per-kernel loop over all devices.  FFTCONV_HOST kernels / outputs are plain host arrays;
 * FFTCONV_DEVICE pointers must live on the device that owns the kernel (fftconv_multi_shard). */
int fftconv_multi_convolve(fftconv_multi *multi, int n_kernel, const float *const *kernels,
                           const int *kernel_h, const int *kernel_w, int kernel_location,
                           float *const *out, int out_location);
/* Block of kernels device `index` (position in devices[]) owns among n_kernel. */
int fftconv_multi_shard(const fftconv_multi *multi, int n_kernel, int index, int *first, int *count);
/* Number of plans, and the plan / device id at a position (for fftconv_plan_get_info, options,
 * or packed device-resident calls on one device; the plan stays owned by the multi handle). */
int fftconv_multi_size(const fftconv_multi *multi);
int fftconv_multi_plan(fftconv_multi *multi, int index, fftconv_plan **plan, int *device);
/* One-shot form: fftconv_convolution_fft with a list of devices instead of one gpu_id. */
int fftconv_convolution_fft_multi(const float *data, int data_h, int data_w, int feature_dim,
                                  int max_kernel_h, int max_kernel_w,
                                  int n_kernel, const float *const *kernels,
                                  const int *kernel_h, const int *kernel_w, const int *kernel_f,
                                  const int *devices, int n_devices,
                                  float *const *out, int *fft_h, int *fft_w);

#ifdef __cplusplus
}
#endif
#endif /* FFTCONV_H */
