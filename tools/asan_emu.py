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
