"""Random shapes for the fp64 systolic engine against the oracle (GPU box): frame sizes around the period / ring-size
boundaries, frame counts around multiples of 64, sweep counts that are no multiple of the slots per pass, thresholds that skip bins.
    python tools/stress_sys64.py [cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lws_amd
from oracle.oracle import Oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
orc = Oracle()
worst, names = 0.0, {}
F_EDGE = [17, 19, 33, 65, 129, 251, 257, 499, 501, 503, 505, 507, 509, 511, 513, 515, 517, 519, 521, 523, 577, 601, 617,
          # 128 frames in flight, two waves per sweep slot (round 5): around its period of 1024 steps and its ring-size boundaries
          619, 621, 641, 769, 1001, 1013, 1015, 1017, 1019, 1021, 1023, 1025, 1027, 1029, 1031, 1033, 1041, 1051, 1061, 1067,
          # 256 frames in flight, four waves per slot
          1069, 1071, 1101, 1501, 2041, 2043, 2045, 2047, 2049, 2051, 2053, 2061, 2081, 2089]
T_EDGE = [1, 2, 3, 5, 57, 58, 59, 61, 63, 64, 65, 66, 67, 121, 122, 123, 125, 127, 128, 129, 130, 131, 200, 255, 256, 257, 258]
for case in range(n_cases):
    Q = int(rng.choice([2, 4]))
    F = int(rng.choice(F_EDGE)) if rng.random() < 0.7 else int(rng.integers(9, 310)) * 2 + 1
    fsize = 2 * (F - 1)
    if fsize % Q:
        F += (Q - fsize % Q) // 2 * 0  # keep F odd; choose a hop that need not divide the frame exactly? it must: adjust frame
        fsize = 2 * (F - 1)
    if fsize % Q:
        continue
    fshift = fsize // Q
    T = int(rng.choice(T_EDGE)) if rng.random() < 0.6 else int(rng.integers(1, 150))
    if 400 < F < 619:
        T = min(T, 70)
    if F >= 619 and T > 140 and rng.random() < 0.7:
        T = min(T, 140)
    iters = int(rng.integers(1, 11))
    alpha = float(rng.choice([1.0, 3.0, 100.0]))
    p = lws_amd.lws(fsize, fshift, batch_iterations=iters, batch_alpha=alpha, precision="fp64")
    mag = np.abs(rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))) * rng.random((T, F)) ** 2
    S = mag * np.exp(2j * np.pi * rng.random((T, F)))
    thr = lws_amd.get_thresholds(iters, alpha, 0.1, 1)
    out = p.batch_lws(S)
    name = p.plan().last_kernel()["name"]
    names[name] = names.get(name, 0) + 1
    ref = orc.batch_lws(S, p.W, thr)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    worst = max(worst, err)
    flag = "" if err < 1e-10 else "   <-- FAIL"
    if flag or case % 10 == 0:
        print("case %3d Q=%d F=%4d T=%4d iters=%2d alpha=%5.1f %-18s err %.2e%s" % (case, Q, F, T, iters, alpha, name, err, flag), flush=True)
    if flag:
        sys.exit(1)
print("worst %.2e over %d cases; kernels: %s" % (worst, n_cases, names))
