import ctypes, sys, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); import util
emu=ctypes.CDLL('/tmp/libfftconv_emu_asan.so')
o=util.Oracle()
def conv(data,mkh,mkw,ks):
    d, kss, n, kp, kh, kw = util.Oracle._prep(data, ks)
    H,W,F=d.shape
    outs=[np.full((util.ceil16(H+mkh-1),util.ceil16(W+mkw-1)),7e7,dtype=np.float32,order='F') for _ in range(n)]
    op=(ctypes.c_void_p*n)(*[x.ctypes.data for x in outs])
    rc=emu.emu_conv_fft(ctypes.c_void_p(d.ctypes.data),H,W,F,mkh,mkw,n,kp,kh,kw,op,None,None)
    assert rc==0, rc
    return outs
shapes=[(64,8,5,10,4,3),(256,256,1,31,31,1),(1024,1024,1,63,63,1),(300,260,1,63,63,2),(40,4096,1,7,127,1),(4096,24,1,127,9,1),(2048,300,1,63,20,1),(512,512,2,31,31,1),(300,4096,2,20,63,1),(200,2048,3,9,63,2),(57,57,1,9,9,1),(33,47,3,7,5,2)]
for mode,grp in [(2,-1),(2,3),(1,-1),(0,-1)]:
    emu.emu_set_tuning(mode,grp)
    for sh in shapes:
        H,W,F,kh,kw,n=sh
        if mode==0 and max(H,W)>1100: continue
        rng=np.random.default_rng(sum(sh))
        data=rng.random((H,W,F),dtype=np.float32); ks=[rng.random((kh,kw,F),dtype=np.float32) for _ in range(n)]
        got=conv(data,kh,kw,ks); ref=o.conv_fft(data,kh,kw,ks)
        err=max(util.rel_err(g,r) for g,r in zip(got,ref))
        assert err<1e-5,(mode,grp,sh,err)
    print('mode',mode,'group',grp,'ok')
# round 4: every new specialised length once along w (row kernels) and once along h (column kernels incl. the 4-column tiles),
# default path mode with the multi-map walk; then the native BASELINE windows under exact_window
emu.emu_set_tuning(2,3)
new=[(24,1300,1,5,40,1),(1300,40,1,30,9,1),(20,1700,2,5,50,1),(1700,40,1,50,9,1),(20,1850,1,3,63,1),(1850,28,1,63,5,1),(14,2200,1,3,70,1),(2200,28,1,70,5,1),
     (16,2500,1,3,60,1),(2500,28,1,50,5,1),(14,2700,1,3,90,1),(2700,28,1,90,5,1),(12,3400,1,3,100,1),(3400,28,1,100,5,1),(12,3750,1,3,60,1),(3750,30,1,60,3,1),
     (10,4400,1,3,127,1),(4400,30,1,127,3,1),(10,5000,1,3,110,1),(5000,30,1,110,3,1),(10,5500,1,3,100,1),(5500,30,1,100,3,1),(10,6000,1,3,100,1),(6000,30,1,90,3,1),
     (10,6900,1,3,127,1),(6900,30,1,127,3,1),(10,7500,1,3,127,1),(7500,30,1,127,3,1),(10,8192,1,3,127,1),(8192,30,1,127,3,1)]
for sh in new:
    H,W,F,kh,kw,n=sh
    rng=np.random.default_rng(sum(sh))
    data=rng.random((H,W,F),dtype=np.float32); ks=[rng.random((kh,kw,F),dtype=np.float32) for _ in range(n)]
    got=conv(data,kh,kw,ks); ref=o.conv_fft(data,kh,kw,ks)
    err=max(util.rel_err(g,r) for g,r in zip(got,ref))
    assert err<1e-5,(sh,err)
print('round-4 lengths ok')
emu.emu_set_exact_window(1)
for sh in [(1024,40,1,63,9,2),(40,1024,2,9,63,1),(1024,1024,1,63,63,1),(4096,28,1,63,5,1),(24,4096,1,5,63,2)]:
    H,W,F,kh,kw,n=sh
    rng=np.random.default_rng(sum(sh))
    data=rng.random((H,W,F),dtype=np.float32); ks=[rng.random((kh,kw,F),dtype=np.float32) for _ in range(n)]
    got=conv(data,kh,kw,ks); ref=o.conv_fft(data,kh,kw,ks)
    err=max(util.rel_err(g,r) for g,r in zip(got,ref))
    assert err<1e-5,(sh,err)
emu.emu_set_exact_window(0)
print('native windows (exact_window) ok')
