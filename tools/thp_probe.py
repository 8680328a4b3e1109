import ctypes, mmap, time, os, threading
print("thp enabled:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
libc = ctypes.CDLL(None, use_errno=True)
libc.mmap.restype = ctypes.c_void_p
libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
N = 1 << 30
def run(huge, threads):
    p = libc.mmap(None, N, 3, 0x22, -1, 0)
    if huge: libc.madvise(p, N, 14)
    t0 = time.perf_counter()
    part = N // threads
    def w(i):
        libc.madvise(ctypes.c_void_p(p + i * part), part, 23)   # MADV_POPULATE_WRITE
    ts = [threading.Thread(target=w, args=(i,)) for i in range(threads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print("populate 1 GiB huge=%d threads=%d: %.1f ms = %.1f GB/s" % (huge, threads, dt * 1e3, N / dt / 1e9))
    libc.munmap(ctypes.c_void_p(p), N)
for huge in (0, 1):
    for th in (1, 4, 8, 16):
        run(huge, th)
