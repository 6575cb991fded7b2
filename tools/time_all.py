import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, lws_amd
from lws_amd import _capi
def t(fsize, fshift, B, T, iters):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift)
    S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(np.complex64)).cuda()
    plan = _capi.Plan(F, p.W); thr = np.zeros(iters)
    plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    print("%-30s %s" % (plan.last_kernel()["name"], " ".join("%.2f" % m for m in ms)), flush=True)
t(1024,256,256,500,100); t(512,128,512,500,100); t(256,64,256,2000,100); t(2048,512,64,6000,60); t(1024,128,256,500,100); t(4096,1024,64,2000,20); t(1024,512,256,500,100); t(1000,250,256,500,100)
