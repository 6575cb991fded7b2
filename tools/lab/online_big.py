import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, lws_amd
from lws_amd import _capi
def run(F, W, S, thr, LA, qdiv, **kw):
    plan = _capi.Plan(F, W[0], W[1], W[2], **kw)
    t0 = time.perf_counter(); out = plan.online(S, thr, LA, qdiv); dt = time.perf_counter() - t0
    name = plan.last_kernel()["name"]; plan.close()
    return out, name, dt
for fsize, fshift, T, iters, LA in ((4096, 1024, 12, 2, 3), (4096, 1024, 40, 10, 3), (3000, 750, 20, 3, 3), (4096, 2048, 16, 3, 2), (4096, 1024, 66, 10, 3), (2048, 512, 30, 10, 5)):
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music")
    rng = np.random.default_rng(T)
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    out, name, dt = run(F, W, S, thr, LA, fsize / fshift)
    gen, name2, dt2 = run(F, W, S, thr, LA, fsize / fshift, force_generic=True)
    err = np.abs(out - gen)
    print(fsize, fshift, "T", T, "it", iters, "LA", LA, name, "%.1f ms" % (dt * 1e3), "vs", name2, "%.1f ms" % (dt2 * 1e3),
          "rel %.2e median %.1e mag %.1e finite %s" % (np.linalg.norm(err) / np.linalg.norm(gen), np.median(err) / np.abs(S).mean(), np.abs(np.abs(out) - np.abs(gen)).max() / np.abs(S).max(), np.isfinite(out).all()), flush=True)
