"""Random shapes for the band engine (lws_band.hip) against the oracle (GPU box): every Q from 2 to 16, stencils of half-width 1 to 10,
hops that divide the frame and hops that do not (general tensors: LWSfractionalQ), frame lengths around the lane periods and ring-size
boundaries, frame counts around multiples of the lanes of a slot, sweep counts that are no multiple of the slots per pass, thresholds that
skip bins and whole passes, fp32 and fp64 plans.  The systolic builds are switched off so that every shape lands on the band engine.
    python tools/stress_band.py [cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LWS_NO_SYSTOLIC"] = "1"
os.environ["LWS_NO_SYS64"] = "1"
import warnings
import numpy as np
import lws_amd
from lws_amd import _capi
from oracle.oracle import Oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
orc = Oracle()
worst32, worst64, names = 0.0, 0.0, {}
T_EDGE = [1, 2, 3, 5, 50, 57, 58, 63, 64, 65, 66, 114, 121, 122, 127, 128, 129, 130, 200, 250, 256, 257]
for case in range(n_cases):
    Q = int(rng.integers(2, 17))
    L = int(rng.choice([1, 3, 5, 5, 5, 6, 7, 8, 9, 10]))
    # frame sizes: multiples of the hop (summarised tensors) or not (general ones)
    general = rng.random() < 0.25 and Q <= 8
    half = int(rng.choice([16, 24, 60, 100, 128, 250, 256, 258, 300, 500, 512, 514, 520, 640, 768, 1000, 1024, 1030]))
    if Q > 8:
        half = min(half, 514)
    fsize = 2 * half
    if general:
        fshift = max(2, int(fsize / (Q - rng.random() * 0.9)))
        if fsize % fshift == 0 or -(-fsize // fshift) != Q:
            general = False
    if not general:
        fsize = (fsize // Q) * Q
        if fsize % 2:
            fsize += Q if Q % 2 else 0
        if fsize % 2 or fsize % Q or fsize < 32:
            continue
        fshift = fsize // Q
    F = fsize // 2 + 1
    if F < 2 * (5 if L <= 5 else 10) + 7 or F % 2 == 0:      # (an even number of bins is refused like in the reference, lws.pyx:223-224)
        continue
    T = int(rng.choice(T_EDGE)) if rng.random() < 0.6 else int(rng.integers(1, 150))
    if F > 600:
        T = min(T, 140)
    iters = int(rng.integers(1, 8))
    alpha = float(rng.choice([1.0, 3.0, 100.0]))
    fp64 = rng.random() < 0.3 and not general
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = lws_amd.lws(fsize, fshift, L=L)
    B = int(rng.integers(1, 4))
    mag = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))) * rng.random((B, T, F)) ** 2
    S = mag * np.exp(2j * np.pi * rng.random((B, T, F))) * (10.0 ** rng.integers(-3, 4, size=(B, 1, 1)))
    thr = lws_amd.get_thresholds(iters, alpha, 0.4, 1)
    plan = _capi.Plan(F, p.W, precision="fp64" if fp64 else "fp32")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = plan.batch(S, thr)
    name = plan.last_kernel()["name"]
    plan.close()
    names[name] = names.get(name, 0) + 1
    if not name.startswith("band"):
        print("case %3d lws(%d,%d) L=%d Q=%d F=%4d T=%4d: %s (not a band shape)" % (case, fsize, fshift, L, p.W.shape[1], F, T, name), flush=True)
        continue
    err = 0.0
    for b in range(B):
        ref = orc.batch_lws(S[b], p.W, thr)
        if fp64:
            err = max(err, np.abs(out[b] - ref).max() / np.abs(ref).max())
        else:
            err = max(err, np.linalg.norm(out[b] - ref) / np.linalg.norm(ref), 1e3 * np.abs(np.abs(out[b]) - np.abs(S[b])).max() / np.abs(S[b]).max())
    bad = err > (1e-9 if fp64 else 2e-3)
    if fp64: worst64 = max(worst64, err)
    else: worst32 = max(worst32, err)
    if bad or case % 10 == 0:
        print("case %3d lws(%d,%d) L=%d Q=%d%s F=%4d T=%4d B=%d iters=%d alpha=%5.1f %-10s err %.2e%s" % (case, fsize, fshift, L, p.W.shape[1], " general" if general else "", F, T, B, iters, alpha, name, err, "   <-- FAIL" if bad else ""), flush=True)
    if bad:
        sys.exit(1)
print("worst fp32 rel-L2 %.2e, fp64 %.2e over %d cases; kernels: %s" % (worst32, worst64, n_cases, names))
