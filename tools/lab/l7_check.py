import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
def chk(fsize, fshift, L, T, n_it, B=2):
    F = fsize//2+1
    p = lws_amd.lws(fsize, fshift, L=L)
    rng = np.random.default_rng(T+n_it)
    S = rng.standard_normal((B,T,F)) + 1j*rng.standard_normal((B,T,F))
    thr = np.linspace(0.6, 0.0, n_it)
    a = _capi.Plan(F, p.W); out = a.batch(S, thr); name = a.last_kernel()["name"]
    g = _capi.Plan(F, p.W, precision="fp64"); ref = g.batch(S, thr); name2 = g.last_kernel()["name"]
    d = np.abs(out-ref)
    bad = np.argwhere(d > 1e-3*np.abs(S).mean())
    print(fsize,fshift,"L",L,"T",T,"it",n_it,name,"vs",name2,"rel %.2e med %.1e" % (np.linalg.norm(out-ref)/np.linalg.norm(ref), np.median(d)/np.abs(S).mean()),"nbad",len(bad), bad[:3].tolist(), flush=True)
for args in ((64,16,7,1,1),(64,16,7,5,2),(64,16,7,70,3),(1024,256,7,40,4),(1024,512,7,70,5),(1000,250,7,33,3),(512,128,6,130,7),(1024,256,6,200,2),(100,25,7,66,4),(1012,253,7,20,3),(1020,255,6,30,3)):
    chk(*args)
B,T,F,iters=256,500,513,20
p = lws_amd.lws(1024,256,L=7)
S = torch.from_numpy((np.random.default_rng(0).standard_normal((B,T,F)) + 0j).astype(np.complex64)).cuda()
for kw in ({}, {"force_generic": True}):
    plan = _capi.Plan(F, p.W, **kw)
    plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize()
    t0=time.perf_counter(); plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(plan.last_kernel()["name"], "%.1f ms %.2f ps/bin-sweep" % (dt*1e3, dt/(B*T*F*iters)*1e12), flush=True)
