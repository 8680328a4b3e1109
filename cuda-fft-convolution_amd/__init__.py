"""Python binding of libfftconv.so, the MI355X-native 2-D FFT-convolution engine.

The host side of the product is C++ (``csrc/fftconv_api.cpp`` and the units beside it (``csrc/plan_internal.hpp`` lists them), the role of the reference's MEX
gateways); this module is only the ctypes stub over the C ABI of ``include/fftconv.h`` plus a
mirror of the reference's MATLAB call surface so tests read like the reference's demo:

    cvcell  = cudaConvolutionFFT(data, maxKH, maxKW, kernelCell[, threads4][, gpuId])
              (reference: src/cudaConvolutionFFT.cu:27-311, demoCudaConvolutionFFT.m:124-129)
    fftData = cudaFFTData(data, kH, kW)                 (src/cudaFFTData.cu:18-160)
    cvcell  = cudaConvFFTData(fftData, kernelCell[, threads4])   (src/cudaConvFFTData.cu:24-306)

Arrays follow MATLAB conventions: ``data`` is H x W x F, kernels are kh x kw x F, every result is
the full FFT_H x FFT_W window (not cropped).  There is no CPU fallback: if the HIP library is
missing or no GPU is present the compute calls raise.

Environment (this stub only; libfftconv.so itself reads none): FFTCONV_LIB = path of an alternate
build of the library (A/B runs, tools/build_variant.sh); FFTCONV_NO_TORCH_RUNTIME = 1 skips the
preload of PyTorch's bundled HIP runtime (see _preload_hip_runtime).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FFTCONV_LIB") or os.path.join(_HERE, "libfftconv.so")  # FFTCONV_LIB: kernel experiments

HOST, DEVICE, AUTO = 0, 1, 2
MEX_ERROR_ID = "cudaConvFFTData:InvalidInput"  # src/cudaConvolutionFFT.cu:30


class FFTConvError(RuntimeError):
    """Raised for any negative status of the C ABI; ``.status`` holds the code and ``.identifier``
    the MEX error id the reference raises for argument errors."""

    def __init__(self, status, message):
        super().__init__("fftconv status %d: %s" % (status, message))
        self.status = status
        self.identifier = MEX_ERROR_ID


class PlanInfo(ctypes.Structure):
    _fields_ = [
        ("data_h", ctypes.c_int), ("data_w", ctypes.c_int), ("feature_dim", ctypes.c_int),
        ("max_kernel_h", ctypes.c_int), ("max_kernel_w", ctypes.c_int),
        ("fft_h", ctypes.c_int), ("fft_w", ctypes.c_int),
        ("transform_h", ctypes.c_int), ("transform_w", ctypes.c_int),
        ("spectrum_rows", ctypes.c_int), ("spectrum_pitch", ctypes.c_int),
        ("gpu_id", ctypes.c_int), ("exact_window", ctypes.c_int),
        ("spectrum_bytes", ctypes.c_size_t), ("map_bytes", ctypes.c_size_t),
        ("workspace_bytes", ctypes.c_size_t),
        ("out_h", ctypes.c_int), ("out_w", ctypes.c_int), ("out_map_bytes", ctypes.c_size_t),
    ]


class PlanOptions(ctypes.Structure):
    """fftconv_plan_options (include/fftconv.h): choices fixed at plan creation."""
    _fields_ = [("struct_size", ctypes.c_size_t), ("kernel_path", ctypes.c_int), ("rows_group", ctypes.c_int),
                ("max_transform", ctypes.c_int), ("exact_window", ctypes.c_int), ("blockwise", ctypes.c_int),
                ("verbose", ctypes.c_int)]

    def __init__(self, kernel_path=0, rows_group=0, max_transform=0, exact_window=0, blockwise=0, verbose=0):
        super().__init__(ctypes.sizeof(PlanOptions), int(kernel_path), int(rows_group), int(max_transform), int(exact_window), int(blockwise),
                         int(verbose))


class CallTiming(ctypes.Structure):
    """fftconv_call_timing: where the time of the last one-shot call of this thread went (ms)."""
    _fields_ = [("plan_ms", ctypes.c_double), ("image_ms", ctypes.c_double), ("convolve_ms", ctypes.c_double),
                ("release_ms", ctypes.c_double), ("total_ms", ctypes.c_double), ("cache_hit", ctypes.c_int)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _options_ptr(options):
    """None, a PlanOptions or a dict of its fields -> (pointer or None, keep-alive)"""
    if options is None:
        return None, None
    if isinstance(options, dict):
        options = PlanOptions(**options)
    return ctypes.byref(options), options


class Profile(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double * 5), ("launches", ctypes.c_long * 5), ("units", ctypes.c_long * 5)]

    NAMES = ("kernel_cols", "spectral_rows", "cols_c2r", "image_cols", "image_rows")

    def as_dict(self):
        return {n: {"ms": self.ms[i], "launches": self.launches[i], "units": self.units[i]}
                for i, n in enumerate(self.NAMES)}


# every symbol include/fftconv.h declares
EXPORTED_SYMBOLS = (
    "fftconv_fft_size16", "fftconv_fft_size_pow2", "fftconv_last_error", "fftconv_version", "fftconv_device_count",
    "fftconv_convolution_fft", "fftconv_convolution_fft_ex",
    "fftconv_cache_configure", "fftconv_cache_clear", "fftconv_cache_stats", "fftconv_last_call_timing",
    "fftconv_plan_create", "fftconv_plan_create_ex",
    "fftconv_plan_is_live", "fftconv_plan_destroy", "fftconv_plan_get_info",
    "fftconv_plan_set_image", "fftconv_plan_spectrum", "fftconv_plan_mark_spectrum_valid",
    "fftconv_plan_use_spectrum_buffer", "fftconv_plan_export_spectrum", "fftconv_plan_import_spectrum",
    "fftconv_plan_convolve", "fftconv_plan_convolve_packed", "fftconv_plan_prepare_kernels_packed",
    "fftconv_plan_synchronize", "fftconv_plan_set_stream",
    "fftconv_plan_set_option", "fftconv_plan_get_option", "fftconv_plan_get_profile", "fftconv_fft_data", "fftconv_conv_fft_data",
    "fftconv_multi_create", "fftconv_multi_destroy", "fftconv_multi_set_image", "fftconv_multi_import_spectrum",
    "fftconv_multi_convolve", "fftconv_multi_set_option", "fftconv_multi_get_option",
    "fftconv_multi_shard", "fftconv_multi_size", "fftconv_multi_plan", "fftconv_convolution_fft_multi",
)

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's); if libfftconv.so pulled in the system copy first, a later
    `import torch` would bring a second runtime into the process and see no GPUs, and streams or
    events could not be shared.  So when torch is installed, its copy is loaded first (by path,
    without importing torch) and libfftconv.so binds to it through the SONAME."""
    if os.environ.get("FFTCONV_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def load_library():
    """Loads libfftconv.so (built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``).
    Fails loudly when it is missing: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFTConvError(-7, "HIP extension %s is missing; build it with `make -C %s`"
                           % (LIB_PATH, os.path.join(_HERE, "csrc")))
    _preload_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    pi = ctypes.POINTER(ctypes.c_int)
    lib.fftconv_fft_size16.argtypes = [ci]
    lib.fftconv_fft_size_pow2.argtypes = [ci]
    lib.fftconv_last_error.restype = ctypes.c_char_p
    lib.fftconv_version.restype = ctypes.c_char_p
    lib.fftconv_device_count.argtypes = [pi]
    lib.fftconv_convolution_fft.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, pi, pi]
    lib.fftconv_convolution_fft_ex.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, vp, pi, pi, vp]
    lib.fftconv_cache_configure.argtypes = [ci, cs]
    lib.fftconv_cache_clear.argtypes = []
    lib.fftconv_cache_stats.argtypes = [ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long),
                                        ctypes.POINTER(cs)]
    lib.fftconv_last_call_timing.argtypes = [ctypes.POINTER(CallTiming)]
    lib.fftconv_plan_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, ci, ci, vp]
    lib.fftconv_plan_create_ex.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, ci, ci, vp, vp]
    lib.fftconv_plan_is_live.argtypes = [vp]
    lib.fftconv_plan_destroy.argtypes = [vp]
    lib.fftconv_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    lib.fftconv_plan_set_image.argtypes = [vp, vp, ci]
    lib.fftconv_plan_spectrum.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(cs)]
    lib.fftconv_plan_mark_spectrum_valid.argtypes = [vp]
    lib.fftconv_plan_use_spectrum_buffer.argtypes = [vp, vp, cs]
    lib.fftconv_plan_export_spectrum.argtypes = [vp, vp, ci]
    lib.fftconv_plan_import_spectrum.argtypes = [vp, vp, ci]
    lib.fftconv_plan_convolve.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci]
    lib.fftconv_plan_convolve_packed.argtypes = [vp, ci, vp, ci, ci, vp]
    lib.fftconv_plan_prepare_kernels_packed.argtypes = [vp, ci, vp, ci, ci]
    lib.fftconv_plan_synchronize.argtypes = [vp]
    lib.fftconv_plan_set_stream.argtypes = [vp, vp]
    lib.fftconv_plan_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_long]
    lib.fftconv_plan_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
    lib.fftconv_plan_get_profile.argtypes = [vp, ctypes.POINTER(Profile), ci]
    lib.fftconv_fft_data.argtypes = [vp, ci, ci, ci, ci, ci, ci, ctypes.POINTER(vp)]
    lib.fftconv_conv_fft_data.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.fftconv_multi_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, ci, ci, pi, ci, vp]
    lib.fftconv_multi_destroy.argtypes = [vp]
    lib.fftconv_multi_set_image.argtypes = [vp, vp, ci]
    lib.fftconv_multi_import_spectrum.argtypes = [vp, vp, ci]
    lib.fftconv_multi_convolve.argtypes = [vp, ci, vp, vp, vp, ci, vp, ci]
    lib.fftconv_multi_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_long]
    lib.fftconv_multi_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]
    lib.fftconv_multi_shard.argtypes = [vp, ci, ci, pi, pi]
    lib.fftconv_multi_size.argtypes = [vp]
    lib.fftconv_multi_plan.argtypes = [vp, ci, ctypes.POINTER(vp), pi]
    lib.fftconv_convolution_fft_multi.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, pi, ci, vp, pi, pi]
    _lib = lib
    # the one-shot entries keep their last few plans (device scratch, host copy threads): released before the
    # interpreter tears the HIP runtime down (the MEX gateways do the same with mexAtExit)
    import atexit
    atexit.register(lib.fftconv_cache_clear)
    return lib


def _check(rc):
    if rc != 0:
        raise FFTConvError(rc, load_library().fftconv_last_error().decode("utf-8", "replace"))


def fft_size16(n):
    """computeFFTsize16 (src/cudaConvFFTData.h:96-102)."""
    return load_library().fftconv_fft_size16(int(n))


def fft_size_pow2(n):
    """computeFFTsize (src/cudaConvFFTData.h:67-94)."""
    return load_library().fftconv_fft_size_pow2(int(n))


def device_count():
    n = ctypes.c_int(0)
    load_library().fftconv_device_count(ctypes.byref(n))
    return n.value


def cache_configure(max_plans=4, max_bytes=0):
    """plan cache of the one-shot entry: plans kept (0 = off), device bytes they may hold (0 = unchanged)"""
    _check(load_library().fftconv_cache_configure(int(max_plans), int(max_bytes)))


def cache_clear():
    _check(load_library().fftconv_cache_clear())


def cache_stats():
    n, h, m, b = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_long(0), ctypes.c_size_t(0)
    _check(load_library().fftconv_cache_stats(ctypes.byref(n), ctypes.byref(h), ctypes.byref(m), ctypes.byref(b)))
    return {"plans": n.value, "hits": h.value, "misses": m.value, "device_bytes": b.value}


def last_call_timing():
    """where the time of this thread's last cudaConvolutionFFT call went (dict of ms + cache_hit)"""
    t = CallTiming()
    _check(load_library().fftconv_last_call_timing(ctypes.byref(t)))
    return t.as_dict()


def _as_matlab_single(a, what):
    """float32, column-major, 3-D (H x W x F); 2-D inputs are H x W x 1 (the reference rejects
    them -- SURVEY D4 -- this engine accepts F = 1)."""
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise FFTConvError(-1, "%s must be single (float32), got %s" % (what, a.dtype))
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3:
        raise FFTConvError(-1, "Invalid %s input" % what)
    return np.asfortranarray(a)


def _kernel_tables(kernels):
    ks = [_as_matlab_single(k, "kernel") for k in kernels]
    n = len(ks)
    ptrs = (ctypes.c_void_p * n)(*[k.ctypes.data for k in ks])
    kh = (ctypes.c_int * n)(*[k.shape[0] for k in ks])
    kw = (ctypes.c_int * n)(*[k.shape[1] for k in ks])
    kf = (ctypes.c_int * n)(*[k.shape[2] for k in ks])
    return ks, ptrs, kh, kw, kf


def _thread_size(threads):
    if threads is None:
        return None, 0, None
    t = np.ascontiguousarray(np.asarray(threads, dtype=np.float64).ravel())
    return ctypes.c_void_p(t.ctypes.data), int(t.size), t


def cudaConvolutionFFT(data, maxKernelH, maxKernelW, kernelCell, threadSize=None, gpuId=0, options=None, out=None):
    """One-shot convolution, host arrays in / host arrays out (list of FFT_H x FFT_W float32,
    Fortran order).  Mirrors the MEX entry of src/cudaConvolutionFFT.cu.  ``options``: a
    PlanOptions (or a dict of its fields) for fftconv_convolution_fft_ex; ``out``: optional list of
    caller buffers (FFT_H x FFT_W float32, Fortran order) to fill instead of fresh arrays."""
    lib = load_library()
    if not isinstance(kernelCell, (list, tuple)):
        raise FFTConvError(-1, "Kernel must be a cell array")  # src/cudaConvolutionFFT.cu:64-65
    d = _as_matlab_single(data, "data")
    H, W, F = d.shape
    ks, kptr, kh, kw, kf = _kernel_tables(kernelCell)
    n = len(ks)
    fh, fw = fft_size16(H + int(maxKernelH) - 1), fft_size16(W + int(maxKernelW) - 1)
    if out is None:
        outs = [np.empty((fh, fw), dtype=np.float32, order="F") for _ in range(n)]
    else:
        outs = list(out)
        if len(outs) != n or any(o.dtype != np.float32 or o.shape != (fh, fw) or not o.flags.f_contiguous for o in outs):
            raise FFTConvError(-1, "out must hold one FFT_H x FFT_W float32 Fortran-order buffer per kernel")
    optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    tptr, tn, _keep = _thread_size(threadSize)
    ofh, ofw = ctypes.c_int(0), ctypes.c_int(0)
    oref, _okeep = _options_ptr(options)
    _check(lib.fftconv_convolution_fft_ex(ctypes.c_void_p(d.ctypes.data), H, W, F, int(maxKernelH), int(maxKernelW),
                                          n, kptr, kh, kw, kf, HOST, tptr, tn, int(gpuId), optr,
                                          ctypes.byref(ofh), ctypes.byref(ofw), oref))
    assert (ofh.value, ofw.value) == (fh, fw)
    return outs


class Plan:
    """Plan API: image spectrum computed once, reused for any number of kernels.  Pointers are
    plain integers (e.g. ``torch.Tensor.data_ptr()``); torch is not a dependency of this module."""

    def __init__(self, H, W, F, maxKernelH, maxKernelW, gpuId=0, stream=0, options=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p(None)
        oref, _okeep = _options_ptr(options)
        _check(self._lib.fftconv_plan_create_ex(ctypes.byref(self._h), int(H), int(W), int(F), int(maxKernelH),
                                                int(maxKernelW), int(gpuId), ctypes.c_void_p(int(stream) or None), oref))
        self.info = PlanInfo()
        _check(self._lib.fftconv_plan_get_info(self._h, ctypes.byref(self.info)))

    # -- lifetime
    def destroy(self):
        if self._h is not None and self._h.value:
            self._lib.fftconv_plan_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    # -- image
    def set_image(self, data):
        """host numpy H x W x F (or H x W) float32"""
        d = _as_matlab_single(data, "data")
        if d.shape != (self.info.data_h, self.info.data_w, self.info.feature_dim):
            raise FFTConvError(-1, "Invalid data input: shape %s does not match the plan" % (d.shape,))
        _check(self._lib.fftconv_plan_set_image(self._h, ctypes.c_void_p(d.ctypes.data), HOST))

    def set_image_device(self, ptr):
        _check(self._lib.fftconv_plan_set_image(self._h, ctypes.c_void_p(int(ptr)), DEVICE))

    def spectrum(self):
        """(device pointer, bytes) of the image spectrum buffer (for the RCCL broadcast)."""
        p, n = ctypes.c_void_p(None), ctypes.c_size_t(0)
        _check(self._lib.fftconv_plan_spectrum(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def export_spectrum(self):
        """the image spectrum in the reference's order (what cudaFFTData returns, src/cudaFFTData.cu:
        90-103): complex64 [F][FFT_W][FFT_H/2+1], unnormalised, == numpy.fft.rfft2 of the padded
        [F][FFT_W][FFT_H] planes.  Needs a plan created with exact_window."""
        i = self.info
        a = np.empty((i.feature_dim, i.fft_w, i.fft_h // 2 + 1), dtype=np.complex64)
        _check(self._lib.fftconv_plan_export_spectrum(self._h, ctypes.c_void_p(a.ctypes.data), HOST))
        return a

    def import_spectrum(self, spectrum):
        """replaces set_image: a spectrum in the reference's order (see export_spectrum)"""
        i = self.info
        a = np.ascontiguousarray(spectrum, dtype=np.complex64)
        if a.shape != (i.feature_dim, i.fft_w, i.fft_h // 2 + 1):
            raise FFTConvError(-1, "spectrum must be [F][FFT_W][FFT_H/2+1], got %s" % (a.shape,))
        _check(self._lib.fftconv_plan_import_spectrum(self._h, ctypes.c_void_p(a.ctypes.data), HOST))

    def use_spectrum_buffer(self, ptr, nbytes):
        """keep the image spectrum in caller-owned device memory (e.g. a torch tensor)"""
        _check(self._lib.fftconv_plan_use_spectrum_buffer(self._h, ctypes.c_void_p(int(ptr) or None), int(nbytes)))

    def mark_spectrum_valid(self):
        _check(self._lib.fftconv_plan_mark_spectrum_valid(self._h))

    # -- convolution
    def convolve(self, kernelCell, out=None):
        """host kernels (list of kh x kw x F float32) -> list of host maps.  ``out``: optional list
        of caller buffers to fill (FFT_H x FFT_W float32, Fortran order; pageable, or pinned for a
        direct DMA copy-out) instead of fresh arrays."""
        ks, kptr, kh, kw, kf = _kernel_tables(kernelCell)
        for k in ks:
            if k.shape[2] != self.info.feature_dim:  # src/cudaConvolutionFFT.cu:242
                raise FFTConvError(-3, "Kernel and Data must have the same number of features and kernel "
                                       "size should be smaller than data size")
        n = len(ks)
        _check(self._lib.fftconv_plan_get_info(self._h, ctypes.byref(self.info)))   # out_h / out_w follow "output_region"
        oh, ow = self.info.out_h, self.info.out_w
        if out is None:
            outs = [np.empty((oh, ow), dtype=np.float32, order="F") for _ in range(n)]
        else:
            outs = list(out)
            if len(outs) != n:
                raise FFTConvError(-1, "out must hold one buffer per kernel")
            for o in outs:
                if (o.dtype != np.float32 or o.shape != (oh, ow)
                        or not o.flags.f_contiguous or not o.flags.writeable):
                    raise FFTConvError(-1, "out buffers must be writable out_h x out_w float32 in Fortran order")
        optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        _check(self._lib.fftconv_plan_convolve(self._h, n, kptr, kh, kw, HOST, optr, HOST))
        return outs

    def convolve_packed_device(self, n, kernels_ptr, kh, kw, out_ptr):
        """n equally sized kernels packed in device memory -> n maps packed in device memory;
        asynchronous on the plan's stream."""
        _check(self._lib.fftconv_plan_convolve_packed(self._h, int(n), ctypes.c_void_p(int(kernels_ptr)),
                                                      int(kh), int(kw), ctypes.c_void_p(int(out_ptr))))

    def prepare_kernels_packed_device(self, n, kernels_ptr, kh, kw):
        """queue the image-independent part of convolve_packed_device (kernel column transforms);
        the next convolve_packed_device call with the same arguments reuses it"""
        _check(self._lib.fftconv_plan_prepare_kernels_packed(self._h, int(n), ctypes.c_void_p(int(kernels_ptr)),
                                                             int(kh), int(kw)))

    def synchronize(self):
        _check(self._lib.fftconv_plan_synchronize(self._h))

    def is_live(self):
        return bool(self._h is not None and self._h.value and self._lib.fftconv_plan_is_live(self._h))

    def set_stream(self, stream):
        """re-bind the plan to another HIP stream (integer handle, 0 = default stream)"""
        _check(self._lib.fftconv_plan_set_stream(self._h, ctypes.c_void_p(int(stream) or None)))

    def set_option(self, name, value):
        _check(self._lib.fftconv_plan_set_option(self._h, name.encode(), int(value)))
        _check(self._lib.fftconv_plan_get_info(self._h, ctypes.byref(self.info)))

    def get_option(self, name):
        v = ctypes.c_long(0)
        _check(self._lib.fftconv_plan_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def profile(self, reset=True):
        pr = Profile()
        _check(self._lib.fftconv_plan_get_profile(self._h, ctypes.byref(pr), 1 if reset else 0))
        return pr.as_dict()


def cudaConvolutionFFTMulti(data, maxKernelH, maxKernelW, kernelCell, gpuIds):
    """One-shot convolution over several GPUs from this one process (fftconv_convolution_fft_multi:
    what src/cudaConvFFTDataStreams.cu set out to be): image transformed on gpuIds[0], spectrum
    peer-copied, kernels dealt in contiguous blocks.  Host arrays in / out like cudaConvolutionFFT."""
    lib = load_library()
    if not isinstance(kernelCell, (list, tuple)):
        raise FFTConvError(-1, "Kernel must be a cell array")
    d = _as_matlab_single(data, "data")
    H, W, F = d.shape
    ks, kptr, kh, kw, kf = _kernel_tables(kernelCell)
    n = len(ks)
    fh, fw = fft_size16(H + int(maxKernelH) - 1), fft_size16(W + int(maxKernelW) - 1)
    outs = [np.empty((fh, fw), dtype=np.float32, order="F") for _ in range(n)]
    optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    devs = (ctypes.c_int * len(gpuIds))(*[int(g) for g in gpuIds])
    ofh, ofw = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib.fftconv_convolution_fft_multi(ctypes.c_void_p(d.ctypes.data), H, W, F, int(maxKernelH), int(maxKernelW),
                                             n, kptr, kh, kw, kf, devs, len(gpuIds), optr, ctypes.byref(ofh), ctypes.byref(ofw)))
    return outs


class MultiPlan:
    """fftconv_multi_*: one plan per listed device driven from this process (image spectrum on
    gpuIds[0], peer copies, contiguous kernel blocks)."""

    def __init__(self, H, W, F, maxKernelH, maxKernelW, gpuIds, options=None):
        self._lib = load_library()
        self._h = ctypes.c_void_p(None)
        devs = (ctypes.c_int * len(gpuIds))(*[int(g) for g in gpuIds])
        oref, _okeep = _options_ptr(options)
        _check(self._lib.fftconv_multi_create(ctypes.byref(self._h), int(H), int(W), int(F), int(maxKernelH), int(maxKernelW),
                                              devs, len(gpuIds), oref))
        # "" or the warning naming destinations without direct peer access to gpuIds[0] (their copies are staged)
        self.warning = (self._lib.fftconv_last_error() or b"").decode()
        self.shape = (int(H), int(W), int(F))
        p, dev = ctypes.c_void_p(None), ctypes.c_int(0)
        _check(self._lib.fftconv_multi_plan(self._h, 0, ctypes.byref(p), ctypes.byref(dev)))
        self.info = PlanInfo()
        _check(self._lib.fftconv_plan_get_info(p, ctypes.byref(self.info)))

    def __len__(self):
        return self._lib.fftconv_multi_size(self._h)

    def set_option(self, name, value):
        """"spectrum_transport" 0 peer copies / 1 one RCCL broadcast; "verbose" (include/fftconv.h)"""
        _check(self._lib.fftconv_multi_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_long(0)
        _check(self._lib.fftconv_multi_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def last_message(self):
        """text of the calling thread's last error / warning (e.g. why the RCCL transport was not used)"""
        return (self._lib.fftconv_last_error() or b"").decode()

    def shard(self, n_kernel, index):
        first, count = ctypes.c_int(0), ctypes.c_int(0)
        _check(self._lib.fftconv_multi_shard(self._h, int(n_kernel), int(index), ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    def set_image(self, data):
        d = _as_matlab_single(data, "data")
        if d.shape != self.shape:
            raise FFTConvError(-1, "Invalid data input: shape %s does not match the plan" % (d.shape,))
        _check(self._lib.fftconv_multi_set_image(self._h, ctypes.c_void_p(d.ctypes.data), HOST))

    def convolve(self, kernelCell):
        ks, kptr, kh, kw, kf = _kernel_tables(kernelCell)
        for k in ks:
            if k.shape[2] != self.shape[2]:
                raise FFTConvError(-3, "Kernel and Data must have the same number of features and kernel "
                                       "size should be smaller than data size")
        n = len(ks)
        outs = [np.empty((self.info.fft_h, self.info.fft_w), dtype=np.float32, order="F") for _ in range(n)]
        optr = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
        _check(self._lib.fftconv_multi_convolve(self._h, n, kptr, kh, kw, HOST, optr, HOST))
        return outs

    def destroy(self):
        if self._h is not None and self._h.value:
            self._lib.fftconv_multi_destroy(self._h)
            self._h = ctypes.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def cudaFFTData(data, kernelH, kernelW, gpuId=0):
    """Two-step API, step 1 (src/cudaFFTData.cu): returns a handle holding the device-resident
    image spectrum (the role of the complex gpuArray the reference returns)."""
    d = _as_matlab_single(data, "data")
    H, W, F = d.shape
    p = Plan(H, W, F, kernelH, kernelW, gpuId)
    p.set_image(d)
    return p


def cudaConvFFTData(fftData, kernelCell, threadSize=None):
    """Two-step API, step 2 (src/cudaConvFFTData.cu)."""
    if not isinstance(fftData, Plan):
        raise FFTConvError(-1, "Invalid input to MEX file.")  # src/cudaConvFFTData.cu:68
    if not isinstance(kernelCell, (list, tuple)):
        raise FFTConvError(-1, "Kernel must be a cell array")
    if threadSize is not None and np.asarray(threadSize).size != 4:  # src/cudaConvFFTData.cu:76-77
        raise FFTConvError(-2, "CUDA Thread Size must be 4 integers")
    return fftData.convolve(kernelCell)
