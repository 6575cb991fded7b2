import numpy as np, torch, lws_amd, time, os
from lws_amd import _capi
def t(fsize, fshift, B, T, iters, precision="fp32", **kw):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, **kw)
    dt = np.complex64 if precision == "fp32" else np.complex128
    S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(dt)).cuda()
    plan = _capi.Plan(F, p.W, precision=precision, force_generic=True); thr = np.zeros(iters)
    plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
    ms = []
    for _ in range(2):
        plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    n = B*T*F*iters
    print("ITEMS=%s %5d/%4d F=%4d %s %-20s %s  ps/bin-sweep %.1f" % (os.environ.get("LWS_GENERIC_ITEMS"), fsize, fshift, F, precision, plan.last_kernel()["name"], " ".join("%.1f" % m for m in ms), min(ms)*1e9/n), flush=True)
t(1024,256,256,500,100); t(1024,256,256,500,100,"fp64"); t(1000,200,256,500,40); t(1024,128,256,500,40)
