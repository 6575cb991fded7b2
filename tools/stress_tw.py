"""Random shapes for the table-twiddle paths (Q = 3, fractional Q): batch sweeps on lws::tw / tw_half / tw_wide, no-future sweeps on
the LDS engine with periodic general weights, online sweeps on k_online4<..., TWT> -- each against the order-exact generic fp32 engine
(same magnitudes on every bin; values while the stage's own amplification of rounding allows: few sweeps / short runs).
usage: PYTHONPATH=. python tools/stress_tw.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import lws_amd
from lws_amd import _capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
cfgs = [(48, 16), (96, 32), (384, 128), (768, 256), (1008, 336), (60, 20), (1020, 340), (996, 332), (984, 328), (1536, 512), (1980, 660),
        (400, 160), (512, 160), (1000, 400), (1024, 384), (600, 250), (80, 32), (1024, 320), (644, 230), (1012, 368), (2048, 768),
        (2000, 800), (2044, 700), (1200, 480), (2048, 640), (160, 64), (320, 128), (800, 320), (240, 96), (1600, 640),
        # five to eight frames per stencil row: the 64-step ring with table twiddles 
        (80, 16), (1000, 200), (960, 192), (96, 16), (768, 128), (1020, 170), (112, 16), (896, 128), (1008, 144), (1024, 160), (1024, 176),
        (1000, 150), (1024, 192), (100, 20), (1012, 184),
        # a hop above half the frame (Q = 2, general weights)
        (512, 300), (1024, 640), (400, 240), (64, 40), (2048, 1280)]
bad = 0
for it in range(cases):
    fs, sh = cfgs[rng.integers(len(cfgs))]
    F = fs // 2 + 1
    L = int(rng.choice([5, 5, 5, 1, 2, 3, 4]))
    kind = str(rng.choice(["batch", "batch", "nofuture", "online"]))
    B = int(rng.integers(1, 4))
    p = lws_amd.lws(fs, sh, L=L, mode="music")
    W = (p.W, p.W_ai, p.W_af)
    zero_phase = rng.random() < 0.25
    scale = 10.0 ** rng.uniform(-3, 3)
    def data(T):
        S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
        return ((np.abs(S) + 0j) if zero_phase else S) * scale
    if kind == "batch":
        T, n = int(rng.integers(1, 160 if fs <= 1100 else 90)), int(rng.integers(1, 12))
        S = data(T)
        thr = np.sort(rng.random(n) * 1.5)[::-1].copy()
        run = lambda pl: pl.batch(S, thr)
        want = "systolic"
    elif kind == "nofuture":
        T, n = int(rng.integers(1, 30)), int(rng.integers(1, 3))
        S = data(T)
        thr = np.sort(rng.random(n))[::-1].copy()
        run = lambda pl: pl.nofuture(S, thr, wsel=_capi.LWS_W_AI)
        want = "nofuture_lds"
    else:
        T, LA, n = int(rng.integers(1, 24)), int(rng.integers(0, 6)), int(rng.integers(1, 5))
        S = data(T)
        thr = lws_amd.get_thresholds(n, 1.0, 0.1, 1)
        run = lambda pl: pl.online(S, thr, LA, fs / sh)
        want = "online_lds"
    fast = _capi.Plan(F, *W)
    a = run(fast); name = fast.last_kernel()["name"]; fast.close()
    gen = _capi.Plan(F, *W, force_generic=True)
    b = run(gen); gen.close()
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)
    mag = np.abs(np.abs(a) - np.abs(b)).max() / np.abs(S).max()
    first = np.linalg.norm(a[:, :6] - b[:, :6]) / np.linalg.norm(b[:, :6])
    ok = name.startswith(want) and mag < 2e-6 and np.isfinite(a).all() and (zero_phase or first < 1e-3) and rel < (0.5 if (zero_phase or kind != "batch") else 6e-3)
    bad += not ok
    print(f"{fs:5d} {sh:4d} L={L} {kind:8s} B={B} T={T:3d} n={n:2d} {name:32s} rel {rel:.1e} first6 {first:.1e} mag {mag:.1e}{'' if ok else '   <<<<<<<<'}", flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
