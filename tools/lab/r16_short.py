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
    print(fsize,fshift,"L",L,"T",T,"it",n_it,name,"vs",name2,"identical" if np.array_equal(out,ref) else "DIFFERENT rel %.2e" % (np.linalg.norm(out-ref)/np.linalg.norm(ref)), flush=True)
for args in ((512,256,5,1,1),(512,256,5,70,3),(512,256,5,200,27),(512,256,3,66,53),(500,250,5,100,30),(256,128,5,5,2),(256,128,5,300,45),(128,64,1,130,90),(252,126,5,64,50),(64,32,5,40,100)):
    chk(*args)
for fs, sh, B, T in ((512,256,512,500),(256,128,512,1000)):
    F=fs//2+1; iters=100
    p = lws_amd.lws(fs,sh)
    S = torch.from_numpy((np.random.default_rng(0).standard_normal((B,T,F)) + 0j).astype(np.complex64)).cuda()
    for env in ("0", "1"):
        os.environ["LWS_SYSTOLIC_NO_R16"] = env
        plan = _capi.Plan(F, p.W)
        plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize()
        ms=[]
        for _ in range(2):
            plan.batch_dev(S.data_ptr(), B, T, np.zeros(iters)); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
        print(plan.last_kernel()["name"], " ".join("%.2f" % m for m in ms), flush=True)
    del os.environ["LWS_SYSTOLIC_NO_R16"]
