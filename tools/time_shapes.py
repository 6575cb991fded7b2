import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time, numpy as np, torch, lws_amd
from lws_amd import _capi
def t(fsize, fshift, B=256, T=500, iters=100):
    F = fsize//2+1
    p = lws_amd.lws(fsize, fshift)
    rng = np.random.default_rng(0)
    S = torch.from_numpy((rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))).astype(np.complex64)).cuda()
    thr = np.zeros(iters)
    plan = _capi.Plan(F, p.W)
    st = torch.cuda.current_stream().cuda_stream
    plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0=time.perf_counter(); plan.batch_dev(S.data_ptr(), B, T, thr, stream=st); torch.cuda.synchronize(); best=min(best,time.perf_counter()-t0)
    n = B*T*F*iters
    print("%5d/%4d F=%4d %-28s %8.2f ms  %.3f ns/bin-sweep x1e3" % (fsize,fshift,F,plan.last_kernel()["name"],best*1e3,best/n*1e12), flush=True)
for fs,fh in ((1024,256),(1000,250),(1012,253),(1020,255),(1004,502),(1024,512)): t(fs,fh)
for fs,fh in ((2048,512),(2004,501),(1032,258)): t(fs,fh,B=256,T=128,iters=30)
