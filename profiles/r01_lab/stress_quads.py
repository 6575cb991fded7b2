# random shapes: systolic kernel vs the generic engine (both fp32), a few sweeps so that rounding differences stay small
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import lws_amd
from lws_amd import _capi
rng = np.random.default_rng(12345)
worst = 0.0
cfgs = [(64, 16), (64, 32), (128, 32), (128, 64), (256, 64), (512, 128), (1024, 256), (1024, 512), (2048, 512), (2048, 1024), (1536, 384)]
for it in range(60):
    fs, sh = cfgs[rng.integers(len(cfgs))]
    F = fs // 2 + 1
    T = int(rng.integers(1, 200))
    B = int(rng.integers(1, 4))
    n = int(rng.integers(1, 16))
    p = lws_amd.lws(fs, sh)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    if rng.random() < 0.3:
        S = np.abs(S) + 0j
    thr = np.sort(rng.random(n) * 1.5)[::-1].copy()
    a = p.plan().batch(S, thr)
    name = p.plan().last_kernel()["name"]
    g = _capi.Plan(F, p.W, force_generic=True)
    b = g.batch(S, thr)
    rel = np.linalg.norm(a - b) / np.linalg.norm(b)
    worst = max(worst, rel)
    flag = "" if rel < 2e-3 else "   <<<<<<<<"
    print(f"{fs:5d} {sh:4d} B={B} T={T:4d} n={n:2d} {name:28s} rel {rel:.2e}{flag}", flush=True)
print("worst", worst)
