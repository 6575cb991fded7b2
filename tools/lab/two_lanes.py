"""can two half-batches of config 3 run side by side on two streams at the speed of one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["LWS_SYSTOLIC_NWG"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import numpy as np, torch, lws_amd
from lws_amd import _capi
B, T, F = 256, 500, 513
pm = lws_amd.lws(1024, 256, mode="music")
W = (pm.W, pm.W_ai, pm.W_af)
thr_nf = lws_amd.get_thresholds(pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma)
thr_on = lws_amd.get_thresholds(pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma)
thr_b = lws_amd.get_thresholds(pm.batch_iterations, pm.batch_alpha, pm.batch_beta, pm.batch_gamma)
g = torch.Generator(device="cuda"); g.manual_seed(1)
mags = torch.sqrt(torch.randn((B, T, F), device="cuda", generator=g) ** 2 + torch.randn((B, T, F), device="cuda", generator=g) ** 2)
def run(plan, st, lo, hi, stream):
    plan.run_dev(st[lo:hi].data_ptr(), hi - lo, T, thr_nf, thr_on, pm.look_ahead, 4.0, thr_b, stream=stream.cuda_stream)
full = _capi.Plan(F, *W); a = _capi.Plan(F, *W); b = _capi.Plan(F, *W)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for label in ("full", "two lanes", "full", "two lanes"):
    st = mags.to(torch.complex64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if label == "full":
        run(full, st, 0, B, s0)
    else:
        run(a, st, 0, B // 2, s1); run(b, st, B // 2, B, s2)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(label, "%.1f ms" % (dt * 1e3), flush=True)
