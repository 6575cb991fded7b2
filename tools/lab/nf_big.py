import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, lws_amd
from lws_amd import _capi
for fsize, fshift, T in ((4096, 1024, 30), (3000, 750, 25), (2048, 512, 40)):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music")
    rng = np.random.default_rng(T)
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    thr = [0.3, 0.0]
    a = _capi.Plan(F, p.W, p.W_ai, p.W_af); t0 = time.perf_counter(); out = a.nofuture(S, thr, wsel=1); dt = time.perf_counter() - t0; name = a.last_kernel()["name"]
    g = _capi.Plan(F, p.W, p.W_ai, p.W_af, force_generic=True); ref = g.nofuture(S, thr, wsel=1); name2 = g.last_kernel()["name"]
    d = np.abs(out - ref)
    print(fsize, fshift, name, "vs", name2, "rel %.2e median %.1e mag %.1e, first frames max %.1e" % (np.linalg.norm(out - ref) / np.linalg.norm(ref), np.median(d) / np.abs(S).mean(),
          np.abs(np.abs(out) - np.abs(ref)).max() / np.abs(S).max(), d[:, :3].max() / np.abs(S).mean()), flush=True)
