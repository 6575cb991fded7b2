"""GPU (-m gpu): fp16-complex storage (LWS_STORAGE_FP16, BASELINE config 5's second leg).

The reference is fp64 only (lwslib.h:6-26), so there is no reference behaviour to match bit for bit: what is pinned is
(i) structure -- same kernels, same schedule, magnitudes returned exactly, untouched bins bit-identical, every
workgroup count giving the same bits --, (ii) VALUES against the storage-rounding model of tests/fp16_model.py (the oracle's fp64
sweeps with state and magnitudes rounded to half where the kernel rounds them) at the fp32 bars of SURVEY 8(c), and (iii) a stated
tolerance against the plain fp64 oracle, next to the fp32 engine's on the same data (the tolerance report of DESIGN.md section 6)."""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def report(out, ref, M):
    d = np.abs(out - ref)
    return {"rel_l2": rel_l2(out, ref), "median": np.median(d) / M.mean(), "p999": np.quantile(d, 0.999) / M.mean(),
            "mag": np.abs(np.abs(out) - M).max() / M.max()}


@pytest.mark.parametrize("fsize,fshift,T", [(64, 16, 40), (128, 32, 130), (1024, 256, 70), (1024, 512, 65), (2048, 512, 50),
                                            (1024, 128, 40), (64, 8, 70), (1000, 250, 70), (2004, 501, 50), (60, 15, 40), (4096, 1024, 40), (512, 128, 70)])
def test_fp16_storage_structure_and_tolerance(oracle, fsize, fshift, T):
    """Complex (random-phase) input: well conditioned, so values can be compared one by one."""
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    S[1] *= 300.0                                # its own scale per spectrogram
    thr = np.array([3.0, 1.2, 0.6, 0.3, 0.1, 0.0, 0.0, 0.0, 0.0])   # threshold 3 x mean: a dropped / nearly empty first sweep
    p16 = lws_amd.lws(fsize, fshift, storage="fp16")
    p32 = lws_amd.lws(fsize, fshift)
    out16 = p16.plan().batch(S, thr)
    name = p16.plan().last_kernel()["name"]
    assert name.startswith("systolic") and name.endswith("_f16"), name
    out32 = p32.plan().batch(S, thr)
    for b in range(2):
        ref = oracle.batch_lws(S[b], p32.W, thr)
        M = np.abs(S[b])
        r16, r32 = report(out16[b], ref, M), report(out32[b], ref, M)
        # magnitudes come from the caller's fp32 targets, not from the fp16 state: as exact as in fp32 storage
        assert r16["mag"] < 2e-6, r16
        # 9 sweeps = 2 passes (narrow), 3 (wide) or 5 (Q = 8): the state was rounded to 11 bits that many times.  The typical bin
        # follows that rounding (median ~ 2^-11); bins whose weighted sum nearly cancels amplify it by 1/|sum| exactly as
        # they amplify fp32 rounding (the fp32 99.9th percentile times 2^13 saturates at the magnitude itself), so the
        # tails are bounded by energy, not pointwise
        assert r16["median"] < 2e-3 and r16["rel_l2"] < 0.3, (r16, r32)
        assert r16["median"] > 20 * r32["median"]          # (and it really is the fp16 path that ran)
        assert r32["rel_l2"] < 3e-3


@pytest.mark.parametrize("fsize,fshift,T,nslots,kernel", [(1024, 256, 150, 7, "systolic_q4_l5_hann_f16"), (2048, 512, 90, 3, "systolic_wide_q4_l5_hann_f16"),
                                                          (1024, 512, 100, 15, "systolic_r16_q2_l5_hann_f16"), (512, 128, 150, 14, "systolic_half_q4_l5_hann_f16"),
                                                          (1024, 128, 90, 2, "systolic_q8_l5_hann_f16")])
def test_fp16_storage_against_its_rounding_model(oracle, fsize, fshift, T, nslots, kernel):
    """Value-level pin of the storage mode: the kernel against tests/fp16_model.py -- the oracle's fp64 sweeps with state and target
    magnitudes rounded to half (nearest even, per-spectrogram power-of-two scale) on the way in and at every pass boundary
    (`nslots` effective sweeps), thresholds compared with the half magnitudes, output = phase of the half state x fp32 magnitude.
    Bars at the fp32 level (the model's arithmetic is fp64, the kernel's fp32: a value that lands on the other side of a half
    rounding boundary differs by 2^-11 in that bin, which is what the 99.9th percentile sees): a truncating conversion, a scale
    that is off by a factor of two or a pass boundary in the wrong place moves the MEDIAN to 1e-4 .. 1e-3."""
    from fp16_model import fp16_storage_batch
    rng = np.random.default_rng(fsize + fshift)
    F = fsize // 2 + 1
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    S[1] *= 300.0                                # its own scale per spectrogram
    thr = np.concatenate([[50.0, 2.0, 0.7], np.zeros(nslots + 3)])          # a dropped sweep, two sparse ones, then dense: 2-3 passes
    p16 = lws_amd.lws(fsize, fshift, storage="fp16")
    out = p16.plan().batch(S, thr)
    assert p16.plan().last_kernel()["name"] == kernel
    for b in range(2):
        model = fp16_storage_batch(oracle, S[b], p16.W, thr, nslots)
        wrong = fp16_storage_batch(oracle, S[b], p16.W, thr, nslots + 1)   # the same model with its pass boundaries one sweep off
        ref = oracle.batch_lws(S[b], p16.W, thr)
        M = np.abs(S[b])
        r, rw, ro = report(out[b], model, M), report(out[b], wrong, M), report(out[b], ref, M)
        # the final write-back quantises both sides to half, so most bins agree to fp32 rounding or differ by a half ulp (2^-11):
        # what tells a right kernel from a wrong one is HOW MANY bins sit on the other side of a rounding boundary
        flips = lambda x: float(np.mean(np.abs(out[b] - x) > 1e-4 * M))
        f, fw, fo = flips(model), flips(wrong), flips(ref)
        print("%s b=%d vs model: rel-L2 %.1e median %.1e p99.9 %.1e, bins off by a half ulp %.4f | model with %d slots: %.4f | fp64 oracle: %.4f (rel-L2 %.1e)"
              % (kernel, b, r["rel_l2"], r["median"], r["p999"], f, nslots + 1, fw, fo, ro["rel_l2"]))
        assert r["rel_l2"] < 1e-3 and r["median"] < 1e-6 and r["p999"] < 1e-3, r      # SURVEY 8(c)'s fp32 bars, against the model
        assert f < 0.02, f                     # (measured: 0.2-0.7 %; a pass boundary one sweep off: 17-31 %; no rounding at all: 85 %)
        assert fw > 5 * f and fo > 5 * f       # the test can tell a pass boundary in the wrong place, and the storage rounding itself


@pytest.mark.parametrize("fsize,fshift,L,T", [(1024, 256, 7, 40), (1000, 250, 4, 40), (1000, 125, 5, 40), (512, 128, 3, 70), (4096, 1024, 2, 20)])
def test_fp16_storage_on_the_other_builds(oracle, fsize, fshift, L, T):
    """fp16 storage with the stencil widths / frame ends / frame sizes that got their systolic builds in round 3 (L = 7, even L,
    Q = 8 with F - 1 = 4 mod 8, short and extra wide frames): same bounds as above."""
    rng = np.random.default_rng(fsize + T + L)
    F = fsize // 2 + 1
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    thr = np.array([1.2, 0.6, 0.3, 0.1, 0.0, 0.0, 0.0])
    p16 = lws_amd.lws(fsize, fshift, L=L, storage="fp16")
    out16 = p16.plan().batch(S, thr)
    name = p16.plan().last_kernel()["name"]
    assert name.startswith("systolic") and name.endswith("_f16"), name
    for b in range(2):
        ref = oracle.batch_lws(S[b], p16.W, thr)
        r16 = report(out16[b], ref, np.abs(S[b]))
        assert r16["mag"] < 2e-6 and r16["median"] < 2e-3 and r16["rel_l2"] < 0.3, r16


def test_never_updated_bins_are_bit_identical_and_noop_schedules():
    rng = np.random.default_rng(3)
    p = lws_amd.lws(1024, 256, storage="fp16")
    S = rng.standard_normal((2, 40, 513)) + 1j * rng.standard_normal((2, 40, 513))
    thr = np.array([1.5, 1.0])                   # only bins above the mean are ever updated
    out = p.plan().batch(S, thr)
    M = np.abs(S)
    quiet = M < 0.98 * M.mean(axis=(1, 2), keepdims=True)     # clear of the fp16 rounding of the comparison
    assert quiet.mean() > 0.3
    assert np.array_equal(out[quiet], S[quiet])
    loud = M > 1.6 * M.mean(axis=(1, 2), keepdims=True)
    assert np.mean(out[loud] != S[loud]) > 0.99
    assert np.array_equal(p.plan().batch(S, [90.0, 40.0]), S)  # every sweep dropped: nothing changes


def test_fp16_workgroup_counts_and_device_io(monkeypatch):
    """Several workgroups per spectrogram hand fp16 rows over through HBM: same bits as one workgroup; and the direct
    device path (complex64 in place) equals the path through the extended buffers."""
    import torch
    rng = np.random.default_rng(11)
    for fsize, fshift, T in ((1024, 256, 260), (2048, 512, 150), (1000, 250, 260), (1012, 253, 200)):
        F = fsize // 2 + 1
        p = lws_amd.lws(fsize, fshift, storage="fp16")
        S = np.abs(rng.standard_normal((3, T, F)) + 1j * rng.standard_normal((3, T, F))).astype(np.complex128)
        thr = lws_amd.get_thresholds(30, 3.0, 0.15, 1)
        monkeypatch.setenv("LWS_SYSTOLIC_NWG", "1")
        ref = p.plan().batch(S, thr)
        for nwg in ("2", "4"):
            monkeypatch.setenv("LWS_SYSTOLIC_NWG", nwg)
            assert np.array_equal(p.plan().batch(S, thr), ref), (fsize, nwg)
        monkeypatch.delenv("LWS_SYSTOLIC_NWG")
        direct = _capi.Plan(F, p.W, storage="fp16")
        padded = _capi.Plan(F, p.W, storage="fp16", direct_io=False)
        S32 = S.astype(np.complex64)
        a, b = torch.from_numpy(S32.copy()).cuda(), torch.from_numpy(S32.copy()).cuda()
        stream = torch.cuda.current_stream().cuda_stream
        direct.batch_dev(a.data_ptr(), 3, T, np.zeros(9), stream=stream)
        padded.batch_dev(b.data_ptr(), 3, T, np.zeros(9), stream=stream)
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert direct.last_kernel()["name"].endswith("_f16")
        assert np.abs(a - b).max() < 2e-6 * np.abs(S32).max()     # (the two output passes normalise in different orders)
        direct.close(); padded.close()


def test_fp16_in_a_pipeline_and_flag_errors():
    """run_lws(mode='music') with fp16 storage: only the batch stage uses it; quality as with fp32 storage."""
    rng = np.random.default_rng(5)
    M = np.abs(rng.standard_normal((120, 513)) + 1j * rng.standard_normal((120, 513)))
    p16 = lws_amd.lws(1024, 256, mode="music", storage="fp16", batch_iterations=60)
    p32 = lws_amd.lws(1024, 256, mode="music", batch_iterations=60)
    o16, o32 = p16.run_lws(M), p32.run_lws(M)
    assert p16.plan().last_kernel()["name"].endswith("_f16")
    assert abs(p16.get_consistency(o16) - p32.get_consistency(o32)) < 0.1
    assert np.abs(np.abs(o16) - M).max() < 2e-6 * M.max()
    with pytest.raises(ValueError):
        _capi.Plan(513, p16.W, precision="fp64", storage="fp16")


def test_config5_crop_tolerance_report(oracle):
    """BASELINE config 5 / SURVEY 8(d): 'fp32 vs fp16-complex storage, tolerance report vs the fp64 restatement on a
    T=2048 crop' -- lws(2048, 512), 200 sweeps of the default schedule, Rayleigh magnitudes with zero phase.  The
    stated bars: fp32 storage as everywhere; fp16 storage: consistency within
    0.3 dB of the fp64 result, magnitudes exact to fp32 rounding, rel-L2 / quantiles reported (zero-phase starts let
    rounding decide individual phases, so they are bounded loosely)."""
    rng = np.random.default_rng(20260928)
    T, F = 2048, 1025
    M = np.abs(rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))).astype(np.float32).astype(np.float64)
    thr = lws_amd.get_thresholds(200, 100, 0.1, 1)
    p32 = lws_amd.lws(2048, 512)
    p16 = lws_amd.lws(2048, 512, storage="fp16")
    ref = oracle.batch_lws(M, p32.W, thr)
    o32 = p32.batch_lws(M, thresholds=thr)
    assert p32.plan().last_kernel()["name"].startswith("systolic_wide_q4")
    o16 = p16.batch_lws(M, thresholds=thr)
    assert p16.plan().last_kernel()["name"] == "systolic_wide_q4_l5_hann_f16"
    c_ref, c32, c16 = (p32.get_consistency(x) for x in (ref, o32, o16))
    r32, r16 = report(o32, ref, M), report(o16, ref, M)
    print("\nconfig-5 crop (2048 x 1025, 200 sweeps) vs fp64 oracle")
    print("  storage   rel-L2      median/mean  99.9pct/mean  max|d|mag|/max  consistency dB (fp64: %.4f)" % c_ref)
    for nm, r, c in (("fp32", r32, c32), ("fp16", r16, c16)):
        print("  %-8s  %.3e   %.3e    %.3e     %.3e       %.4f" % (nm, r["rel_l2"], r["median"], r["p999"], r["mag"], c))
    # fp32 storage: SURVEY 8(c)'s bars; the 99.9th percentile is given 3x the room here (200 sweeps from a zero-phase
    # start on 2.1 M bins: a handful more near-cancelling bins than in the 500 x 513 x 100 case the bar was set on)
    assert r32["rel_l2"] < 1e-3 and r32["median"] < 1e-6 and r32["p999"] < 3e-3 and abs(c32 - c_ref) < 0.05 and r32["mag"] < 1e-6
    # fp16 storage: same quality (consistency), exact magnitudes; pointwise the typical bin carries the 2^-11 rounding
    # of ~67 passes, the tail is phase flips of ill-conditioned bins (bounded in energy)
    assert r16["mag"] < 1e-6 and abs(c16 - c_ref) < 0.3, (c16, c_ref, r16)
    assert r16["median"] < 1e-2 and r16["rel_l2"] < 0.5, r16
