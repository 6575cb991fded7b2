"""Random shapes for the fp64 online and no-future stages on their LDS engines (lws_online64.hip, lws_nofuture.hip in double): each
result must equal the order-exact generic engine's BIT FOR BIT (every sum is taken in its order), whatever the frame length, hop,
L, look-ahead, number of frames, batch size, scale of the data or kind of start (complex / magnitudes only).  Shapes the engines
do not take (rows that do not fit the LDS, Q outside {2,3,4,8}) must fall back to generic_fp64 -- also checked, by name.
usage: PYTHONPATH=. python tools/stress_fp64_stages.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import lws_amd
from lws_amd import _capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
cfgs = [(64, 16), (64, 32), (64, 8), (48, 16), (128, 32), (256, 64), (512, 128), (512, 256), (512, 64), (1024, 256), (1024, 512), (1024, 128),
        (768, 256), (1000, 250), (1012, 253), (1020, 255), (60, 15), (100, 25), (300, 75), (420, 105), (516, 129), (96, 32), (384, 128),
        (1008, 336), (1536, 384), (1536, 512), (2048, 512), (2048, 1024), (2048, 256), (1100, 275), (76, 38), (252, 126),
        # not served: fall back to the generic engine
        (1000, 200), (1024, 160), (1024, 384), (4096, 1024)]
bad, served = 0, 0
for it in range(cases):
    fs, sh = cfgs[rng.integers(len(cfgs))]
    F = fs // 2 + 1
    L = int(rng.choice([5, 5, 5, 5, 1, 2, 3, 4]))
    kind = str(rng.choice(["online", "online", "nofuture"]))
    B = int(rng.integers(1, 5))
    p = lws_amd.lws(fs, sh, L=L, mode="music", precision="fp64")
    W = (p.W, p.W_ai, p.W_af)
    zero_phase = rng.random() < 0.4
    scales = 10.0 ** rng.uniform(-3, 3, size=(B, 1, 1))
    T = int(rng.integers(1, 40 if fs <= 1100 else 20))
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    S = ((np.abs(S) + 0j) if zero_phase else S) * scales
    if kind == "online":
        LA, n = int(rng.integers(0, 8)), int(rng.integers(1, 6))
        thr = lws_amd.get_thresholds(n, 1.0, 0.1, 1)
        run = lambda pl: pl.online(S, thr, LA, fs / sh)
    else:
        LA, n = 0, int(rng.integers(1, 3))
        thr = np.sort(rng.random(n))[::-1].copy()
        wsel = int(rng.choice([_capi.LWS_W_AI, _capi.LWS_W_AI, _capi.LWS_W]))
        run = lambda pl: pl.nofuture(S, thr, wsel=wsel)
    fast = _capi.Plan(F, *W, precision="fp64")
    a = run(fast); name = fast.last_kernel()["name"]; fast.close()
    gen = _capi.Plan(F, *W, precision="fp64", force_generic=True)
    b = run(gen); gname = gen.last_kernel()["name"]; gen.close()
    same = np.array_equal(a, b)
    lds = name != "generic_fp64"
    served += lds
    ok = same and np.isfinite(a).all() and gname == "generic_fp64"
    bad += not ok
    print(f"{fs:5d} {sh:4d} L={L} {kind:8s} B={B} T={T:3d} LA={LA} n={n} {name:28s} {'bits equal' if same else 'DIFFERENT  <<<<<<<<'}", flush=True)
print("cases", cases, "on the LDS engines", served, "failures", bad)
sys.exit(1 if bad else 0)
