#!/usr/bin/env python3
"""What tolerance does BASELINE config 3 (run_lws of mode='music': no-future -> online -> batch on 500 x 513) reach in fp32?
Stage by stage against the fp64 oracle, for the production (LDS) engines and for the order-exact generic fp32 engine, each
stage measured twice: CHAINED (the engine's own previous stage as input: what a caller gets) and ISOLATED (the oracle's fp64
result of the previous stage as input: what this stage alone adds).  With the shipped NoFuture_LWSQ4 addressing (compat) and with
the anyQ semantics (compat off).  Prints one JSON object; tests/test_gpu_parity.py asserts bars derived from it.
    PYTHONPATH=. python tools/config3_tolerance.py [--T 500]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lws_amd
from oracle.oracle import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--seed", type=int, default=20260928 + 3)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
M = np.abs(rng.standard_normal((a.T, 513)) + 1j * rng.standard_normal((a.T, 513))).astype(np.float32).astype(np.float64)
mean = M.mean()
orc = Oracle()


def metrics(x, ref):
    d = np.abs(x - ref)
    out = {"rel_l2": float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), "median": float(np.median(d) / mean),
           "p999": float(np.quantile(d, 0.999) / mean), "frac_gt_1e-3": float(np.mean(d > 1e-3 * mean))}
    for n in (8, 16, 32, 64):      # the first frames alone: before the stage's own sensitivity has amplified anything
        out["rel_l2_first%d" % n] = float(np.linalg.norm(x[:n] - ref[:n]) / np.linalg.norm(ref[:n]))
    return out


res = {}
for compat in (True, False):
    kw = dict(mode="music", nofuture_q4_compat=compat)
    p = lws_amd.lws(1024, 256, **kw)
    pg = lws_amd.lws(1024, 256, force_generic=True, **kw)
    thr_nf = lws_amd.get_thresholds(p.nofuture_iterations, p.nofuture_alpha, p.nofuture_beta, p.nofuture_gamma)
    thr_on = lws_amd.get_thresholds(p.online_iterations, p.online_alpha, p.online_beta, p.online_gamma)
    thr_b = lws_amd.get_thresholds(p.batch_iterations, p.batch_alpha, p.batch_beta, p.batch_gamma)
    r0 = orc.nofuture_lws(M, p.W_ai, thr_nf, compat=compat)
    r1 = orc.online_lws(r0, p.W, p.W_ai, p.W_af, thr_on, p.look_ahead, p.fshift)
    r2 = orc.batch_lws(r1, p.W, thr_b)
    blk = {"consistency_oracle": [float(p.get_consistency(r)) for r in (r0, r1, r2)]}
    # the reference's own sensitivity: the same fp64 arithmetic on a stage input rounded to complex64 (one fp32 ulp)
    r1p = orc.online_lws(r0.astype(np.complex64).astype(np.complex128), p.W, p.W_ai, p.W_af, thr_on, p.look_ahead, p.fshift)
    blk["oracle_on_c64_rounded_input"] = {"online": metrics(r1p, r1)}
    for name, eng in (("lds", p), ("generic", pg)):
        c0 = eng.nofuture_lws(M); k0 = eng.plan().last_kernel()["name"]
        c1 = eng.online_lws(c0); k1 = eng.plan().last_kernel()["name"]
        c2 = eng.batch_lws(c1); k2 = eng.plan().last_kernel()["name"]
        i1 = eng.online_lws(r0)
        i2 = eng.batch_lws(r1)
        blk[name] = {"kernels": [k0, k1, k2],
                     "chained": {"nofuture": metrics(c0, r0), "online": metrics(c1, r1), "batch": metrics(c2, r2)},
                     "isolated": {"online": metrics(i1, r1), "batch": metrics(i2, r2)},
                     "consistency": [float(eng.get_consistency(c)) for c in (c0, c1, c2)],
                     "max_rel_magnitude_error": float(np.abs(np.abs(c2) - M).max() / M.max())}
        if name == "generic":
            blk["lds_vs_generic"] = {"nofuture": metrics(cl[0], c0), "online": metrics(cl[1], c1), "batch": metrics(cl[2], c2)}
        cl = (c0, c1, c2)
    res["compat" if compat else "anyq"] = blk
print(json.dumps(res, indent=1))
