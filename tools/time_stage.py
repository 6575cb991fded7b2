"""Time ONE stage of run_lws(mode='music') on the device, repeated, for A/B runs and for profiler passes around a single kernel.
    PYTHONPATH=. python tools/time_stage.py --stage online [--fsize 1024 --fshift 256 --B 256 --T 500 --reps 3]
LWS_HIP_LIB=... selects a kernel-variant build of the library (ctypes binding)."""
import argparse, time
import numpy as np, torch
import lws_amd

ap = argparse.ArgumentParser()
ap.add_argument("--stage", default="online", choices=["nofuture", "online", "batch"])
ap.add_argument("--fsize", type=int, default=1024)
ap.add_argument("--fshift", type=int, default=256)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--LA", type=int, default=None)
ap.add_argument("--iters", type=int, default=None)
ap.add_argument("--precision", default="fp32", choices=["fp32", "fp64"])
ap.add_argument("--L", type=int, default=5)
a = ap.parse_args()
F = a.fsize // 2 + 1
pm = lws_amd.lws(a.fsize, a.fshift, L=a.L, mode="music", precision=a.precision)
plan = pm.plan()
g = torch.Generator(device="cuda"); g.manual_seed(1)
re = torch.randn((a.B, a.T, F), device="cuda", generator=g); im = torch.randn((a.B, a.T, F), device="cuda", generator=g)
mags = torch.sqrt(re * re + im * im); del re, im
cdt = torch.complex128 if a.precision == "fp64" else torch.complex64
state = torch.empty((a.B, a.T, F), dtype=cdt, device="cuda")
thr_nf = lws_amd.get_thresholds(pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma)
it_on = a.iters or pm.online_iterations
thr_on = lws_amd.get_thresholds(it_on, pm.online_alpha, pm.online_beta, pm.online_gamma)
thr_b = lws_amd.get_thresholds(a.iters or pm.batch_iterations, pm.batch_alpha, pm.batch_beta, pm.batch_gamma)
LA = pm.look_ahead if a.LA is None else a.LA
state.copy_(mags)
plan.nofuture_dev(state.data_ptr(), a.B, a.T, thr_nf, wsel=1)
start = state.clone()   # what the online stage starts from
for rep in range(a.reps):
    state.copy_(mags if a.stage == "nofuture" else start)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if a.stage == "nofuture": plan.nofuture_dev(state.data_ptr(), a.B, a.T, thr_nf, wsel=1)
    elif a.stage == "online": plan.online_dev(state.data_ptr(), a.B, a.T, thr_on, LA, a.fsize / a.fshift)
    else: plan.batch_dev(state.data_ptr(), a.B, a.T, thr_b)
    info = plan.last_kernel()
    torch.cuda.synchronize()
    print("%s %dx%dx%d LA=%d: %s kernel %.3f ms wall %.3f ms  max|d mag| %.2e" % (a.stage, a.B, a.T, F, LA, info["name"], info["ms"], 1e3 * (time.perf_counter() - t0),
          float((state.abs() - mags).abs().max())), flush=True)
