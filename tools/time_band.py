"""Per-bin time of the band engine (lws_band.hip) on the shapes it was built for, beside the generic engine and -- on shapes both
take -- the systolic builds.  PYTHONPATH=. python tools/time_band.py [quick]"""
import os
import sys
import numpy as np, torch, lws_amd
from lws_amd import _capi
quick = len(sys.argv) > 1 and sys.argv[1] in ("quick", "q8w")
one = len(sys.argv) > 1 and sys.argv[1] == "q8w"
def t(fsize, fshift, B, T, iters, precision="fp32", force_generic=False, reps=3, default=False, **kw):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, **kw)
    dt = np.complex64 if precision == "fp32" else np.complex128
    S = torch.from_numpy(np.abs(np.random.default_rng(0).standard_normal((B, T, F))).astype(dt)).cuda()
    plan = _capi.Plan(F, p.W, precision=precision, force_generic=force_generic); thr = lws_amd.get_thresholds(iters, 100, 0.1, 1) if default else np.zeros(iters)     # default: the reference's default schedule (first ~38 of 100 sweeps are no-ops)
    plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        plan.batch_dev(S.data_ptr(), B, T, thr); torch.cuda.synchronize(); ms.append(plan.last_kernel()["ms"])
    n = B * T * F * iters
    print("%5d/%4d L=%2d F=%4d B=%3d T=%4d it=%3d %-6s %-28s %s  ps/bin-sweep %.2f" % (fsize, fshift, kw.get("L", 5), F, B, T, iters, precision, plan.last_kernel()["name"],
          " ".join("%.2f" % m for m in ms), min(ms) * 1e9 / n), flush=True)
    plan.close()
t(2048, 256, 256, 250, 20, reps=1 if one else 3)
if one:
    sys.exit(0)
t(2048, 256, 256, 500, 20)
t(2048, 256, 256, 500, 100, default=True, reps=1)
t(1024, 64, 256, 500, 10)
t(1024, 256, 256, 500, 40, L=8)
if quick:
    os.environ["LWS_BAND_NO_HELPERS"] = "1"      # the exact builds with one wave per slot, for comparison
    t(2048, 256, 256, 500, 20); t(1024, 64, 256, 500, 10); t(1024, 256, 256, 500, 40, L=8)
    del os.environ["LWS_BAND_NO_HELPERS"]
if not quick:
    t(1024, 256, 256, 500, 40, L=7); t(1024, 256, 256, 500, 20, L=10); t(2000, 400, 256, 250, 20); t(2048, 320, 256, 250, 20); t(8192, 2048, 64, 250, 20)
    t(2048, 256, 256, 250, 10, precision="fp64"); t(1024, 128, 256, 500, 10, precision="fp64"); t(1024, 64, 256, 500, 4, precision="fp64"); t(768, 256, 256, 500, 20, precision="fp64")
    # the generic engine on the same plans
    t(2048, 256, 64, 250, 4, force_generic=True, reps=1); t(1024, 64, 64, 500, 2, force_generic=True, reps=1); t(1024, 256, 64, 500, 8, force_generic=True, reps=1, L=8)
    t(1024, 128, 64, 500, 2, precision="fp64", force_generic=True, reps=1)
    # shapes the systolic builds take, on the band engine
    os.environ["LWS_NO_SYSTOLIC"] = "1"
    t(1024, 256, 256, 500, 40); t(1024, 128, 256, 500, 20); t(2048, 512, 256, 250, 40); t(1000, 200, 256, 500, 20)
