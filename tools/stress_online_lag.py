"""Random shapes of the online stage: the default (smallest, often odd) lag between sweeps against LWS_ONLINE_EVEN_LAG=1, bit for bit.
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
    subprocess.check_call([sys.executable, __file__, "run", "odd", n], env=env)
    env["LWS_ONLINE_EVEN_LAG"] = "1"
    subprocess.check_call([sys.executable, __file__, "run", "even", n], env=env)
    a, b = np.load("/tmp/lws_lag_odd.npz"), np.load("/tmp/lws_lag_even.npz")
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    print("cases", len(a.files), "differing", len(bad), bad[:10])
