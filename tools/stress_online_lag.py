"""Random shapes of the online stage: the default (smallest) lag between sweeps against LWS_ONLINE_LAG_PLUS=1, 2, 3, bit for bit.
    python tools/stress_online_lag.py [cases]     (GPU; tests/test_gpu_online_lag.py is the short version)"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "run":
    import lws_amd
    tag, n = sys.argv[2], int(sys.argv[3])
    rng = np.random.default_rng(7)
    out = {}
    for i in range(n):
        N, hop = [(64, 16), (128, 32), (256, 64), (512, 128), (1024, 256), (2048, 512), (256, 128), (1024, 512),
                  (768, 256), (1000, 200), (960, 160), (1024, 384), (400, 160), (600, 150)][int(rng.integers(0, 14))]
        T = int(rng.integers(1, 60)); B = int(rng.integers(1, 4)); LA = int(rng.integers(0, 6)); nit = int(rng.integers(1, 12))
        p = lws_amd.lws(N, hop, mode="music", online_iterations=nit, look_ahead=LA)
        S = rng.rayleigh(1.0, (B, T, N // 2 + 1)).astype(np.complex128)
        out[f"c{i}"] = np.asarray(p.online_lws(S))
        if i == 0: print(tag, p.plan().last_kernel())
    np.savez(f"/tmp/lws_lag_{tag}.npz", **out)
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "200"
    env = dict(os.environ)
    env["LWS_ONLINE_LAYOUT"] = "4"      # (the launcher weighs the layouts by their lags: keep it from changing engines)
    env.pop("LWS_ONLINE_LAG_PLUS", None)
    subprocess.check_call([sys.executable, __file__, "run", "plus0", n], env=env)
    a = np.load("/tmp/lws_lag_plus0.npz")
    for plus in (1, 2, 3):
        env["LWS_ONLINE_LAG_PLUS"] = str(plus)
        subprocess.check_call([sys.executable, __file__, "run", "plus%d" % plus, n], env=env)
        b = np.load("/tmp/lws_lag_plus%d.npz" % plus)
        bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
        print("lag +%d: cases" % plus, len(a.files), "differing", len(bad), bad[:10])
