import sys, time, numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import util
fc = util.load_package(); orc = util.Oracle()
print(fc.load_library().fftconv_version(), "devices", fc.device_count())
rng = np.random.default_rng(0)
cases = [(64,8,5,10,4,3),(33,47,3,7,5,2),(256,256,1,31,31,1),(100,90,2,13,17,2),(1,1,1,1,1,1),(5,5,2,5,5,1),(1,40,2,1,9,2),(1024,1024,1,63,63,2),(300,200,3,21,9,3)]
for (H,W,F,kh,kw,n) in cases:
    data = rng.random((H,W,F),dtype=np.float32)
    ks = [rng.random((kh,kw,F),dtype=np.float32) for _ in range(n)]
    if n>1: ks[1] = rng.random((max(1,kh-2),max(1,kw-1),F),dtype=np.float32)
    o = orc.conv_fft(data,kh,kw,ks)
    t=time.time(); g = fc.cudaConvolutionFFT(data,kh,kw,ks); t=time.time()-t
    err = max(util.rel_err(a,b) for a,b in zip(g,o))
    print((H,W,F,kh,kw,n), "fft", g[0].shape, "err %.2e"%err, "t %.3f"%t, "" if err<1e-5 else "<<<< BAD", flush=True)
