"""end to end (stft -> run_lws -> consistency / istft) on the frame sizes that got new kernels in round 3"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, lws_amd
rng = np.random.default_rng(0)
x = rng.standard_normal(16000 * 4)
for fsize, fshift, kw in ((512, 128, {}), (256, 64, {}), (1000, 250, {}), (4096, 1024, {}), (1024, 256, dict(L=7)), (1000, 125, {}), (1024, 256, dict(L=3)),
                          (512, 128, dict(mode="music")), (1000, 250, dict(mode="music")), (1024, 256, dict(mode="music", L=3)), (4096, 1024, dict(mode="music"))):
    p = lws_amd.lws(fsize, fshift, **kw)
    X = p.stft(x)
    M = np.abs(X)
    c0 = p.get_consistency(M.astype(np.complex128))
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t0 = time.perf_counter(); Y = p.run_lws(M); dt = time.perf_counter() - t0
    c1 = p.get_consistency(Y)
    y = p.istft(Y)
    ok = np.isfinite(Y).all() and np.abs(np.abs(Y) - M).max() < 1e-5 * M.max() and c1 > c0 + 3
    print("lws(%d,%d,%s) %s frames x %d bins: consistency %.2f -> %.2f dB, %.1f ms, last kernel %s%s%s" % (fsize, fshift, kw, X.shape[0], X.shape[1], c0, c1, dt * 1e3,
          p.plan().last_kernel()["name"], "  [warned: generic]" if any("generic" in str(q.message) for q in w) else "", "" if ok else "   <<<<<< CHECK"), flush=True)
