import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
def t(fsize, fshift, B, T, iters, **kw):
    F = fsize//2+1
    p = lws_amd.lws(fsize, fshift)
    rng = np.random.default_rng(0)
    S = torch.from_numpy((rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))).astype(np.complex64)).cuda()
    thr = np.zeros(iters)
    plan = _capi.Plan(F, p.W, **kw)
    st = torch.cuda.current_stream().cuda_stream
    plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0=time.perf_counter(); plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    n = B*T*F*iters
    print("%5d/%4d F=%4d B=%d T=%d it=%d %-28s %8.2f ms  %.3f ps/bin-sweep" % (fsize,fshift,F,B,T,iters,plan.last_kernel()["name"],best*1e3,best/n*1e12), flush=True)
t(1024,256,256,500,100)
t(512,128,256,1000,100)
t(512,128,512,500,100)
t(256,64,256,2000,100)
t(512,256,256,1000,100)
t(1032,258,256,128,30)
t(1032,258,256,128,30,force_generic=True)
t(1000,250,256,128,30,force_generic=True)
t(1000,250,256,128,30)
