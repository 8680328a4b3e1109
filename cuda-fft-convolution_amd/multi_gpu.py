"""Filter sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference's intent (src/cudaConvFFTDataStreams.cu:273-328,338-447): one image spectrum,
kernels dealt out over the devices, the spectrum copied from GPU 0 to the others
(cudaMemcpyPeerAsync, :279-289).  Here: contiguous filter blocks per rank and ONE broadcast of
the spectrum buffer (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" in the
CPU tests).  Outputs stay sharded -- no reduce or gather is needed, maps are independent.

The functions are written against a small engine protocol so the same orchestration drives the
HIP plan on GPUs (bench.py) and the host emulator in the world_size-2 gloo tests:
    engine.compute_spectrum(spec_tensor)       rank 0: image -> spectrum, in place in spec_tensor
    engine.convolve(spec_tensor, first, count) this rank's filters [first, first+count)
"""


def filter_shard(n_filters, rank, world):
    """Contiguous block of filters owned by `rank`: (first, count).  Blocks differ by at most one
    filter; ranks beyond n_filters get (n_filters, 0)."""
    if world < 1 or not (0 <= rank < world) or n_filters < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_filters, world)
    first = rank * base + min(rank, extra)
    count = base + (1 if rank < extra else 0)
    return first, count


def sharded_convolution(engine, spec, n_filters, rank, world, dist=None, src=0):
    """One step of the multi-GPU hot path.  `spec` is this rank's spectrum buffer (a tensor the
    communication backend can broadcast).  Returns whatever engine.convolve returns for the shard."""
    if rank == src:
        engine.compute_spectrum(spec)
    if world > 1:
        dist.broadcast(spec, src=src)      # the single collective of the path
    first, count = filter_shard(n_filters, rank, world)
    return engine.convolve(spec, first, count)
