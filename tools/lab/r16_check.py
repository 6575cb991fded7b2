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
    os.environ["LWS_SYSTOLIC_NO_R16"] = "1"
    g = _capi.Plan(F, p.W); ref = g.batch(S, thr); name2 = g.last_kernel()["name"]
    del os.environ["LWS_SYSTOLIC_NO_R16"]
    d = np.abs(out-ref)
    print(fsize,fshift,"L",L,"T",T,"it",n_it,name,"vs",name2,"identical" if np.array_equal(out,ref) else "rel %.2e max %.1e" % (np.linalg.norm(out-ref)/np.linalg.norm(ref), d.max()), flush=True)
for args in ((1024,512,5,1,1),(1024,512,5,5,2),(1024,512,5,70,3),(1024,512,5,200,16),(1024,512,5,40,31),(1000,500,5,33,17),(1024,512,3,66,9),(1024,512,1,130,20),(800,400,5,100,45),(1024,512,4,50,15),(1012,506,5,64,5)):
    chk(*args)
B,T,F,iters=256,500,513,100
p = lws_amd.lws(1024,512)
S = torch.from_numpy((np.random.default_rng(0).standard_normal((B,T,F)) + 0j).astype(np.complex64)).cuda()
for env in ("0", "1"):
    os.environ["LWS_SYSTOLIC_NO_R16"] = env
    plan = _capi.Plan(F, p.W)
    plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize()
    ms=[]
    for _ in range(3):
        plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    print(plan.last_kernel()["name"], " ".join("%.2f" % m for m in ms), flush=True)
for args in ((2048,1024,5,1,1),(2048,1024,5,70,3),(2048,1024,5,140,20),(2048,1024,3,200,9),(2004,1002,5,66,15),(1100,550,5,130,8),(2048,1024,4,257,22)):
    chk(*args)
B,T,F,iters=64,6000,1025,60
p = lws_amd.lws(2048,1024)
S = torch.from_numpy((np.random.default_rng(0).standard_normal((B,T,F)) + 0j).astype(np.complex64)).cuda()
for env in ("0", "1"):
    os.environ["LWS_SYSTOLIC_NO_R16"] = env
    plan = _capi.Plan(F, p.W)
    plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize()
    ms=[]
    for _ in range(2):
        plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    print(plan.last_kernel()["name"], " ".join("%.2f" % m for m in ms), flush=True)
