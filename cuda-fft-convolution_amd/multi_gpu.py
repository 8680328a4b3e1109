"""The N > 1 steps of the hot path, one process per GPU (torch.distributed): THE implementation,
used by bench.py, by the world_size-2 gloo CPU test (host emulator as engine) and by the -m gpu
test (two processes on one GPU, HIP plans as engines).

Filter sharding (BASELINE configs[3]; the reference's intent, src/cudaConvFFTDataStreams.cu:
273-328 per-GPU plans, :279-289 spectrum copy GPU 0 -> GPU g, :338-447 kernels dealt over the
plans, :452-468 barrier): rank r owns a contiguous block of the filters, rank `src` transforms
the image, ONE broadcast of the spectrum buffer (RCCL over xGMI through torch.distributed's
"nccl" backend; "gloo" in the tests) is the only collective; maps stay sharded.

Image streaming (BASELINE configs[4]; async H2D / D2H intent of src/cudaConvFFTDataStreams.cu:
368-371,429-430): every rank convolves its own images with all kernels, the H2D copy of image
i + 1 on a side stream behind the compute of image i; no collective.

Both are written against a small engine protocol so the same orchestration drives HIP plans and
the host emulator:
    engine.sync                              stream / event helper (NullSync on the CPU)
    engine.new_spectrum()                    a buffer the communication backend can broadcast
    engine.compute_spectrum(spec, image)     image -> spectrum, in place in `spec` (current stream)
    engine.prepare_kernels(first, count)     optional image-independent part of convolve
    engine.convolve(spec, first, count)      this rank's filters [first, first + count) -> maps
    engine.new_image_buffer() / engine.upload(buf, host_image)      (image streaming only)
"""
import contextlib
import time


def filter_shard(n_filters, rank, world):
    """Contiguous block of filters owned by `rank`: (first, count).  Blocks differ by at most one
    filter; ranks beyond n_filters get (n_filters, 0)."""
    if world < 1 or not (0 <= rank < world) or n_filters < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_filters, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def image_shard(n_images, rank, world):
    """Contiguous block of images owned by `rank` (image streaming): (first, count)."""
    return filter_shard(n_images, rank, world)


class NullSync:
    """Stream / event helper of an engine that computes synchronously on the host."""

    def event(self):
        return None

    def side(self):
        return contextlib.nullcontext()

    def record(self, ev, side=False):
        pass

    def wait(self, ev, side=False):
        pass


class TorchStreamSync:
    """Two HIP streams of one device through torch: `main` (the plan's stream, where the maps are
    computed) and `side` (image transform + broadcast of the next step, or H2D copies)."""

    def __init__(self, torch, device, main=None):
        self.torch = torch
        self.main = main if main is not None else torch.cuda.current_stream(device)
        self.side_stream = torch.cuda.Stream(device)

    def event(self):
        return self.torch.cuda.Event()

    def side(self):
        return self.torch.cuda.stream(self.side_stream)

    def record(self, ev, side=False):
        ev.record(self.side_stream if side else self.main)

    def wait(self, ev, side=False):
        # an event that was never recorded is complete: the first use of every buffer does not wait
        (self.side_stream if side else self.main).wait_event(ev)


class FilterShardedConvolver:
    """One image against n_filters kernels sharded over the ranks.

        submit(image)   rank src: image -> spectrum buffer b; all ranks: broadcast of buffer b
                        (side stream: runs beside the maps of the previous step)
        convolve()      this rank's filter block against the oldest submitted spectrum
        step(image)     submit + convolve

    `depth` spectrum buffers: with depth = 2 the caller may submit step k + 1 before it convolves
    step k, which takes the image transform and the broadcast off the critical path of every step
    but the first.  A buffer is overwritten only after the convolve that read it (events)."""

    def __init__(self, engine, dist, rank, world, n_filters, src=0, depth=2, always_collective=False, time_broadcast=None,
                 side_work=None):
        """time_broadcast: None, "wall" (host clock around the call: backends that block the host, gloo) or "event"
        (an event pair on the side stream around it: nccl); broadcast_ms() returns what was measured.
        side_work: callable(stream handle) queued on the side stream where the broadcast goes -- bench.py --contend puts
        a stand-in for the collective's kernel there on a one-GPU box (tools/microbench/cu_hog)."""
        self.engine, self.dist, self.rank, self.world, self.src = engine, dist, rank, world, src
        self.time_broadcast = time_broadcast
        self.side_work = side_work
        # (the step shares the GPU with the collective's kernels: the plan's persistent column kernels take their tiles from
        #  a queue -- plan option "dynamic_tiles", on by default from 864-point transforms on -- so a CU the collective holds delays nobody's share)
        self._bc_wall, self._bc_events = [], []
        self.first, self.count = filter_shard(n_filters, rank, world)
        self.depth = max(1, int(depth))
        self.collective = world > 1 or always_collective
        self.spec = [engine.new_spectrum() for _ in range(self.depth)]
        s = engine.sync
        self.ready = [s.event() for _ in range(self.depth)]      # spectrum b complete on this rank
        self.consumed = [s.event() for _ in range(self.depth)]   # the convolve that read buffer b is done
        self.n_sub = self.n_conv = 0
        self._prepared = -1          # step whose kernels were prepared

    def _prepare(self):
        """image-independent part of the next convolve (the kernels' column transforms): the engine
        may run it beside whatever is queued after it -- the image transform or the broadcast"""
        prep = getattr(self.engine, "prepare_kernels", None)
        if prep is not None and self._prepared != self.n_conv:
            prep(self.first, self.count)
            self._prepared = self.n_conv

    def submit(self, image=None):
        if self.n_sub - self.n_conv >= self.depth:
            raise RuntimeError("submit: all %d spectrum buffers hold steps that were not convolved yet" % self.depth)
        if self.n_sub == self.n_conv:
            self._prepare()          # nothing in flight: ahead of this step's own image transform
        b = self.n_sub % self.depth
        s = self.engine.sync
        with s.side():
            s.wait(self.consumed[b], side=True)
            if self.rank == self.src:
                self.engine.compute_spectrum(self.spec[b], image)
            if self.collective:
                # the single collective of the path; issued from the side stream: the backend orders
                # it behind the transform above and wait() orders the side stream behind it
                t0 = ev0 = None
                if self.time_broadcast == "wall":
                    t0 = time.perf_counter()
                elif self.time_broadcast == "event":
                    torch = self.engine.torch
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record(torch.cuda.current_stream())
                work = self.dist.broadcast(self.spec[b], src=self.src, async_op=True)
                work.wait()
                if t0 is not None:
                    self._bc_wall.append((time.perf_counter() - t0) * 1e3)
                elif ev0 is not None:
                    ev1.record(torch.cuda.current_stream())
                    self._bc_events.append((ev0, ev1))
            if self.side_work is not None:
                self.side_work(self.engine.torch.cuda.current_stream().cuda_stream)
            s.record(self.ready[b], side=True)
        self.n_sub += 1

    def broadcast_ms(self, reset=True):
        """milliseconds of every broadcast timed since the last reset (device events are synchronised here)"""
        out = list(self._bc_wall)
        for ev0, ev1 in self._bc_events:
            ev1.synchronize()
            out.append(ev0.elapsed_time(ev1))
        if reset:
            self._bc_wall, self._bc_events = [], []
        return out

    def convolve(self):
        if self.n_conv >= self.n_sub:
            raise RuntimeError("convolve: nothing submitted")
        b = self.n_conv % self.depth
        s = self.engine.sync
        self._prepare()                       # (pipelined steps: here, beside the broadcast)
        s.wait(self.ready[b])
        res = self.engine.convolve(self.spec[b], self.first, self.count)
        s.record(self.consumed[b])
        self.n_conv += 1
        return res

    def step(self, image=None):
        self.submit(image)
        return self.convolve()

    def run(self, images, on_result=None):
        """Convolves a sequence of images (each against this rank's filter block), the image
        transform + broadcast of image k + 1 overlapped with the maps of image k.  On ranks other
        than src the entries of `images` are ignored (None is fine)."""
        images = list(images)
        last = None
        if images:
            self.submit(images[0])
        for k in range(len(images)):
            if k + 1 < len(images) and self.depth > 1:
                self.submit(images[k + 1])
            last = self.convolve()
            if on_result is not None:
                on_result(k, last)
            if k + 1 < len(images) and self.depth == 1:
                self.submit(images[k + 1])
        return last


class ImageStreamedConvolver:
    """This rank's images against all kernels, H2D of image i + 1 (side stream, two device
    buffers) behind the compute of image i.  No collective: ranks are independent."""

    def __init__(self, engine, n_filters, time_uploads=False):
        self.engine, self.n_filters = engine, n_filters
        self.time_uploads = time_uploads       # an event pair on the side stream around every image's H2D copy (upload_ms())
        self._up_events = []
        self.spec = engine.new_spectrum()
        self.buf = [engine.new_image_buffer() for _ in range(2)]
        s = engine.sync
        self.copied = [s.event() for _ in range(2)]
        self.consumed = [s.event() for _ in range(2)]   # the image transform that read buffer b is done
        self.n_run = 0          # images convolved so far (over all run() calls): image i uses buffer i & 1
        self.last_buf = None    # index of the device buffer holding the image convolved last

    def _upload(self, b, host_image):
        s = self.engine.sync
        with s.side():
            s.wait(self.consumed[b], side=True)     # also across run() calls: the buffer's last reader
            ev0 = None
            if self.time_uploads:
                torch = self.engine.torch
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream())
            self.engine.upload(self.buf[b], host_image)
            if ev0 is not None:
                ev1.record(torch.cuda.current_stream())
                self._up_events.append((ev0, ev1))
            s.record(self.copied[b], side=True)

    def upload_ms(self, reset=True):
        out = []
        for ev0, ev1 in self._up_events:
            ev1.synchronize()
            out.append(ev0.elapsed_time(ev1))
        if reset:
            self._up_events = []
        return out

    def run(self, host_images, on_result=None):
        host_images = list(host_images)
        s = self.engine.sync
        last = None
        base = self.n_run & 1       # continue the alternation across calls (the events order the reuse either way)
        if host_images:
            self._upload(base, host_images[0])
        for i in range(len(host_images)):
            b = (base + i) & 1
            if i + 1 < len(host_images):
                self._upload(1 - b, host_images[i + 1])   # next image's H2D while this one is convolved
            prep = getattr(self.engine, "prepare_kernels", None)
            if prep is not None:
                prep(0, self.n_filters)            # image-independent: may run beside the image transform
            s.wait(self.copied[b])
            self.engine.compute_spectrum(self.spec, self.buf[b])
            s.record(self.consumed[b])
            last = self.engine.convolve(self.spec, 0, self.n_filters)
            self.last_buf = b
            if on_result is not None:
                on_result(i, last)
        self.n_run += len(host_images)
        return last


class HipPlanEngine:
    """Engine over one fftconv Plan (HIP kernels) with device-resident kernels and maps, through
    torch tensors for memory and streams.  `kernels` is this rank's block, packed [n][F][kw][kh];
    convolve() returns the device tensor [n][FFT_W][FFT_H] it fills (reused by every call).

    `kernels_host` (a pinned host tensor of the same shape): the kernels are uploaded INSIDE every step, as the
    reference's per-kernel loop does (cudaMemcpy of every kernel, src/cudaConvolutionFFT.cu:221-222; SURVEY 8(d) counts
    it in the timed region) -- into two device buffers in turn, on an upload stream of their own, so that the copy for
    step k + 1 runs beside the maps of step k (events order buffer reuse behind the convolve that last read it)."""

    def __init__(self, torch, fc, plan, device, kernels, kh, kw, first=0, main_stream=None, overlap=True, out=None,
                 defer_prepare=False, kernels_host=None):
        self.torch, self.fc, self.plan, self.device = torch, fc, plan, device
        self.kernels, self.kh, self.kw, self.first = kernels, kh, kw, first
        self.count = int(kernels.shape[0])
        # the stream the plan is bound to (where the maps are computed)
        self.main_stream = main_stream if main_stream is not None else torch.cuda.current_stream(device)
        # overlap: a side stream for the next step's image transform / broadcast / H2D copies;
        # without it everything is queued on the plan's stream in program order (no events needed)
        self.sync = TorchStreamSync(torch, device, self.main_stream) if overlap else NullSync()
        info = plan.info
        # `out`: reuse another engine's map buffer (bench.py times a second, default-options plan into the same maps)
        self.out = out if out is not None else torch.empty((max(1, self.count), info.fft_w, info.fft_h), dtype=torch.float32, device=device)
        self._keep = None
        # defer_prepare: prepare_kernels only records its request and the kernels' column pass rides in the launch of
        # the image transform that follows ON THE PLAN'S STREAM (one launch fewer per step: single-GPU steps without a
        # side stream, image streaming).  Where the transform runs on a side stream or on another rank (filter
        # sharding) the pass must be queued at once, ahead of the wait for the spectrum: the default.
        plan.set_option("defer_prepare", 1 if defer_prepare else 0)
        self.kernels_host = kernels_host if (kernels_host is not None and self.count) else None
        self.uploads = 0
        # small kernel sets (<= 512 KiB: a few microseconds of PCIe) are not copied at all: the kernels' column pass reads the
        # pinned host memory itself (pinned memory is mapped into the device's address space), so the step still moves
        # its kernels over PCIe but pays no copy launch and no cross-stream wait (two waits cost cfg1 / cfg2 ~13 us a step);
        # larger sets take the upload stream, which hides the copy behind the previous step
        self.zero_copy = self.kernels_host is not None and self.kernels_host.numel() * self.kernels_host.element_size() <= (512 << 10)
        if self.kernels_host is not None and not self.zero_copy:
            self.kbuf = [kernels, torch.empty_like(kernels)]
            self.up_stream = torch.cuda.Stream(device)
            self.k_uploaded = [torch.cuda.Event(), torch.cuda.Event()]
            self.k_consumed = [torch.cuda.Event(), torch.cuda.Event()]
            self.k_cur = 0
            self.k_waited = False
            self._upload(0)

    # -- kernels uploaded per step
    def _upload(self, b):
        with self.torch.cuda.stream(self.up_stream):
            self.up_stream.wait_event(self.k_consumed[b])      # never recorded = complete
            self.kbuf[b].copy_(self.kernels_host, non_blocking=True)
            self.k_uploaded[b].record(self.up_stream)
        self.uploads += 1

    def _kernels_ptr(self):
        """device pointer of this step's kernels (ordering the plan's stream behind their upload once per step)"""
        if self.kernels_host is None:
            return self.kernels.data_ptr()
        if self.zero_copy:
            return self.kernels_host.data_ptr()
        if not self.k_waited:
            self.main_stream.wait_event(self.k_uploaded[self.k_cur])
            self.k_waited = True
        return self.kbuf[self.k_cur].data_ptr()

    def new_spectrum(self):
        return self.torch.empty(self.plan.info.spectrum_bytes, dtype=self.torch.uint8, device=self.device)

    def new_image_buffer(self):
        i = self.plan.info
        return self.torch.empty((i.feature_dim, i.data_w, i.data_h), dtype=self.torch.float32, device=self.device)

    def upload(self, buf, host_image):
        buf.copy_(host_image, non_blocking=True)    # pinned host tensor [F][W][H]: asynchronous H2D

    def compute_spectrum(self, spec, image):
        """image: device tensor [F][W][H].  Runs on torch's CURRENT stream (the convolver makes the
        side stream current for it): the plan is re-bound for the call."""
        cur = self.torch.cuda.current_stream(self.device)
        self.plan.use_spectrum_buffer(spec.data_ptr(), spec.numel())
        rebind = cur.cuda_stream != self.main_stream.cuda_stream
        if rebind:
            self.plan.set_stream(cur.cuda_stream)
        try:
            self.plan.set_image_device(image.data_ptr())
        finally:
            if rebind:
                self.plan.set_stream(self.main_stream.cuda_stream)
        self._keep = image

    def prepare_kernels(self, first, count):
        assert (first, count) == (self.first, self.count)
        if count:
            self.plan.prepare_kernels_packed_device(count, self._kernels_ptr(), self.kh, self.kw)

    def convolve(self, spec, first, count):
        assert (first, count) == (self.first, self.count), "engine was built for another filter block"
        self.plan.use_spectrum_buffer(spec.data_ptr(), spec.numel())
        self.plan.mark_spectrum_valid()
        if count:
            self.plan.convolve_packed_device(count, self._kernels_ptr(), self.kh, self.kw, self.out.data_ptr())
            if self.zero_copy:
                self.uploads += 1                  # (read over PCIe by the kernels' column pass of this step)
            elif self.kernels_host is not None:    # this step's buffer is free once the convolve has read it; the next step's upload starts now
                self.k_consumed[self.k_cur].record(self.main_stream)
                self.k_cur ^= 1
                self.k_waited = False
                self._upload(self.k_cur)
        return self.out
