"""The team engine (lws_team.hip) against the generic engine: online / no-future stage times of shapes no LDS engine takes.
    PYTHONPATH=. python tools/time_team.py [--generic]      (--generic also times the generic engine: minutes)"""
import sys, time
import numpy as np, torch
import lws_amd
from lws_amd import _capi
CASES = [(1024, 64, 5, 64, 500), (1024, 64, 5, 256, 500), (1024, 256, 8, 64, 500), (1200, 100, 5, 64, 500), (2048, 128, 5, 32, 300), (1024, 112, 5, 64, 300)]
for fsize, fshift, L, B, T in CASES:
    F = fsize // 2 + 1
    pm = lws_amd.lws(fsize, fshift, L=L, mode="music")
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    re = torch.randn((B, T, F), device="cuda", generator=g); im = torch.randn((B, T, F), device="cuda", generator=g)
    mags = torch.sqrt(re * re + im * im); del re, im
    thr_nf = lws_amd.get_thresholds(pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma)
    thr_on = lws_amd.get_thresholds(pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma)
    for force in ([False, True] if "--generic" in sys.argv else [False]):
        plan = _capi.Plan(F, pm.W, pm.W_ai, pm.W_af, force_generic=force)
        state = torch.empty((B, T, F), dtype=torch.complex64, device="cuda")
        state.copy_(mags)
        plan.nofuture_dev(state.data_ptr(), B, T, thr_nf, wsel=1); torch.cuda.synchronize()
        nf = plan.last_kernel()
        plan.online_dev(state.data_ptr(), B, T, thr_on, pm.look_ahead, fsize / fshift); torch.cuda.synchronize()
        on = plan.last_kernel()
        print("lws(%d,%d,L=%d) %dx%dx%d: no-future %s %.1f ms | online %s %.1f ms  max|d mag| %.1e" % (
            fsize, fshift, L, B, T, F, nf["name"], nf["ms"], on["name"], on["ms"], float((state.abs() - mags).abs().max())), flush=True)
        plan.close()
