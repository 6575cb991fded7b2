"""Per-bin time of the table-twiddle builds beside the static ones (config 2's volume, dense sweeps).  PYTHONPATH=. python tools/time_tw.py"""
import os
import numpy as np, torch, lws_amd
from lws_amd import _capi
def t(fsize, fshift, B, T, iters, **kw):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, **kw)
    S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(np.complex64)).cuda()
    plan = _capi.Plan(F, p.W); thr = np.zeros(iters)
    plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    n = B * T * F * iters
    print("%5d/%4d F=%4d %-32s %s  ps/bin-sweep %.2f" % (fsize, fshift, F, plan.last_kernel()["name"], " ".join("%.2f" % m for m in ms), min(ms) * 1e9 / n), flush=True)
t(1024, 256, 256, 500, 100); t(768, 256, 256, 500, 100); t(1000, 400, 256, 500, 100); t(1024, 384, 256, 500, 100)
t(400, 160, 512, 500, 100); t(512, 160, 512, 500, 100); t(384, 128, 512, 500, 100); t(512, 128, 512, 500, 100)
t(2048, 768, 64, 2000, 30); t(2048, 512, 64, 2000, 30)
t(1024, 128, 256, 500, 40); t(1000, 200, 256, 500, 40); t(768, 128, 256, 500, 40); t(896, 128, 256, 500, 40); t(1024, 160, 256, 500, 40)
os.environ["LWS_SYSTOLIC_NO_TW"] = "1"
t(768, 256, 64, 500, 20); t(400, 160, 128, 500, 20); t(2048, 768, 16, 500, 10); t(1000, 200, 64, 500, 10); t(1024, 160, 64, 500, 10)
