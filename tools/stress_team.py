"""Random shapes for the team engine (lws_team.hip) on the GPU box: every Q from 2 to 20, stencils of half-width 1 to 10, hops that divide
the frame and hops that do not (general tensors), frames from 17 to 601 bins, 1 to 70 frames, look-ahead 0 to 6, fp32 and fp64 plans.
Bit for bit against the order-exact generic engine: the order-exact kernel (k_team_online_ordered), and the re-associating kernels with ONE
lane per bin (LWS_TEAM_LANES=1: their schedule, ring, placement and zero row with the generic engine's order of terms) -- online sweeps
with the window in LDS and in memory, no-future sweeps; then the production team size against the generic engine at the short-run bars.
    python tools/stress_team.py [cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LWS_TEAM_FIRST"] = "1"          # shapes the LDS engines would take land on the team engine too
import warnings
import numpy as np
import lws_amd
from lws_amd import _capi

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def env(**kw):
    for k in ("LWS_TEAM_LANES", "LWS_TEAM_NO_RING", "LWS_TEAM_FP64", "LWS_TEAM_ORDERED", "LWS_TEAM_DBG_POISON", "LWS_TEAM_NCH3"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


done, worst, names, meds = 0, 0.0, {}, []
while done < n_cases:
    Q = int(rng.integers(2, 21))
    L = int(rng.choice([1, 2, 3, 5, 5, 5, 6, 8, 10]))
    half = int(rng.choice([16, 20, 36, 64, 100, 128, 150, 256, 300, 384, 512, 600]))
    fsize = 2 * half
    if rng.random() < 0.3:
        fshift = max(2, int(fsize / (Q - rng.random() * 0.9)))
    else:
        fsize = max(2 * Q, (fsize // (2 * Q)) * 2 * Q)
        fshift = fsize // Q
    F = fsize // 2 + 1
    if fshift < 2 or F < 17 or F % 2 == 0 or F < 2 * L + 3 or -(-fsize // fshift) > 20:
        continue
    T = int(rng.choice([1, 2, 3, 4, 7, 12, 25, 40, 70]))
    LA = int(rng.integers(0, 7))
    iters = int(rng.integers(1, 5))
    fp64 = rng.random() < 0.5
    prec = "fp64" if fp64 else "fp32"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            p = lws_amd.lws(fsize, fshift, L=L, mode="music")
        except ValueError:
            continue
    B = int(rng.integers(1, 3))
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    if rng.random() < 0.5:
        S = np.abs(S).astype(complex)               # zero phase: what run_lws feeds
    S *= 10.0 ** rng.integers(-3, 4, size=(B, 1, 1))
    thr = lws_amd.get_thresholds(iters, float(rng.choice([1.0, 3.0])), 0.4, 1)
    qdiv = fsize / fshift
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env()
        gen = _capi.Plan(F, p.W, p.W_ai, p.W_af, precision=prec, force_generic=True)
        ref_on = gen.online(S, thr, LA, qdiv)
        ref_nf = gen.nofuture(S, thr, wsel=1)
        gen.close()
        plan = _capi.Plan(F, p.W, p.W_ai, p.W_af, precision=prec)
        tag = "lws(%d,%d,L=%d) Q=%d %s F=%d T=%d LA=%d it=%d B=%d" % (fsize, fshift, L, p.Q, prec, F, T, LA, iters, B)
        # the order-exact kernel
        env(LWS_TEAM_ORDERED=1)
        out = plan.online(S, thr, LA, qdiv)
        name = plan.last_kernel()["name"]
        names[name] = names.get(name, 0) + 1
        if name.startswith("team_online_ordered"):
            assert np.array_equal(out, ref_on), ("ordered", tag, np.abs(out - ref_on).max())
        # one lane per bin on the re-associating kernels: window in LDS, in memory; no-future
        for kw in (dict(LWS_TEAM_LANES=1, LWS_TEAM_FP64=1), dict(LWS_TEAM_LANES=1, LWS_TEAM_FP64=1, LWS_TEAM_NCH3=1), dict(LWS_TEAM_LANES=1, LWS_TEAM_FP64=1, LWS_TEAM_NO_RING=1)):
            env(**kw)
            out = plan.online(S, thr, LA, qdiv)
            name = plan.last_kernel()["name"]
            if name.startswith("team_online") and not np.array_equal(out, ref_on):
                d = np.abs(out - ref_on)
                bad = np.argwhere(d > 0)
                print("MISMATCH", kw, tag, "max", d.max(), "nbad", len(bad), "first", bad[:5].tolist(), "last", bad[-3:].tolist(), "zero phase", bool(np.all(S.imag == 0)), "thr", thr, flush=True)
                for trial_kw in (kw, dict(kw, LWS_TEAM_DBG_POISON=1), dict(kw, LWS_TEAM_NO_RING=1)):
                    env(**trial_kw)
                    res = [plan.online(S, thr, LA, qdiv) for _ in range(10)]
                    print("  ", trial_kw, "bad %d/10" % sum(not np.array_equal(r_, ref_on) for r_ in res), "with NaN %d" % sum(bool(np.isnan(r_).any()) for r_ in res), flush=True)
                np.savez("gpurun_out/stress_team_fail.npz", S=S, thr=thr, ref=ref_on, out=out, meta=np.array([fsize, fshift, L, T, LA, iters, B, int(fp64)]))
                raise AssertionError(("one lane", kw, tag))
        env(LWS_TEAM_LANES=1, LWS_TEAM_FP64=1)
        out = plan.nofuture(S, thr, wsel=1)
        name = plan.last_kernel()["name"]
        if name.startswith("team_nofuture"):
            assert np.array_equal(out, ref_nf), ("one lane, no-future", tag, np.abs(out - ref_nf).max())
        # production team size: the generic engine's magnitudes; its values on the first frames
        env(LWS_TEAM_FP64=1)
        out = plan.online(S, thr, LA, qdiv)
        name = plan.last_kernel()["name"]
        names[name] = names.get(name, 0) + 1
        if name.startswith("team_online"):
            scale = np.abs(S).max(axis=(1, 2), keepdims=True)
            assert (np.abs(np.abs(out) - np.abs(ref_on)) / scale).max() < (1e-12 if fp64 else 3e-6), ("magnitudes", tag)
            n = min(T, 4)
            d = (np.abs(out[:, :n] - ref_on[:, :n]) / scale).max()
            worst = max(worst, d if not fp64 else 0.0)
            # (no bar on the values: a bin whose sum nearly cancels takes its phase from rounding, and the recursion amplifies that within the
            #  look-ahead's frames -- the bit-for-bit checks above are the test; the median difference is reported)
            meds.append(float(np.median(np.abs(out[:, :n] - ref_on[:, :n]) / scale)))
            # ... and it must not depend on what the LDS held before: the same bits with the allocation and its tail poisoned
            env(LWS_TEAM_FP64=1, LWS_TEAM_DBG_POISON=1)
            again = plan.online(S, thr, LA, qdiv)
            assert np.array_equal(again, out), ("production team size, poisoned LDS", tag, np.abs(again - out).max(), bool(np.isnan(again).any()))
        plan.close()
    done += 1
    if done % 20 == 0:
        print(done, "cases;", names, "worst fp32 first-frames difference %.2e" % worst, flush=True)
print("stress_team: %d shapes clean; kernels %r; production team size vs the generic engine, first frames: median of medians %.1e" % (done, names, float(np.median(meds)) if meds else 0.0))
