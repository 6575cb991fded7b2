"""The fp64 systolic batch engine (lws_amd/csrc/lws_sys64.hip) against the oracle and against the order-exact generic engine.

The engine takes a bin's sum in another order than lwslib.cpp:297-354 (scatter form), so the bar is rounding, not bits:
<= 1e-11 of the largest value on random-phase input (observed: 1e-15 .. 1e-14), <= 1e-10 from a zero-phase start (real-valued
input, the documented usage; observed <= 5e-13).  The zero-phase start is the delicate one: the DC and Nyquist bins of the reference
stay EXACTLY real (its pairwise cancellation is exact), and that line is unstable -- an engine that keeps them real to rounding only
drifts off it within three sweeps (tools/zero_phase_sensitivity.py; the oracle itself does when its input gets 1e-16 rad of phase).
The kernel takes the imaginary part of these two sums from the k = 0 taps alone (lws_sys64.hip: Wave::step) and stays on it.
"""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def _spec(rng, T, F, real=False):
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    return np.abs(S).astype(complex) if real else S


@pytest.fixture(scope="module")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


CASES = [
    # fsize, fshift, T, iters          what it exercises
    (64, 16, 9, 5),                    # Q = 4, fewer frames than lanes
    (64, 16, 70, 7),                   # two blocks of 64 frames (lanes wrap to the next block), 3 passes
    (64, 32, 67, 6),                   # Q = 2
    (256, 64, 130, 4),                 # 129 bins
    (1024, 256, 20, 4),                # 513 bins: frame period 528 (16 steps more than 64 lanes x 8)
    (1024, 512, 75, 9),                # Q = 2 at 513 bins
    (996, 249, 66, 4),                 # 499 bins, not a multiple of anything
    (1200, 300, 10, 4),                # 601 bins: one sweep slot on 64 lanes -> the 128-lane geometry (round 5)
    # 128 frames in flight, two waves per sweep slot (round 5: frames of ~620 to ~1070 bins)
    (2048, 512, 20, 5),                # 1025 bins, Q = 4: two slots per pass, fewer frames than lanes
    (2048, 512, 140, 3),               # lanes wrap to the next block of 128 frames
    (2048, 1024, 131, 6),              # Q = 2 at 1025 bins: four slots per pass, 6 sweeps = two passes
    (1536, 384, 30, 4),                # 769 bins: the period is the 128 lanes' own (no surplus steps)
    (2100, 525, 12, 2),                # 1051 bins: one sweep slot only
    # 256 frames in flight, four waves per sweep slot (frames of ~1070 to ~2090 bins)
    (4096, 1024, 20, 3),               # 2049 bins, Q = 4: one slot per pass
    (4096, 2048, 270, 4),              # Q = 2: two slots per pass, lanes wrap to the next block of 256 frames
    (3000, 750, 40, 2),                # 1501 bins
    (2200, 550, 30, 2),                # 1101 bins: the first frames past the 128-lane geometry
]


@pytest.mark.parametrize("fsize,fshift,T,iters", CASES)
def test_against_oracle(fsize, fshift, T, iters, oracle):
    rng = np.random.default_rng(fsize + T)
    p = lws_amd.lws(fsize, fshift, batch_iterations=iters, batch_alpha=1.0, precision="fp64")
    F = fsize // 2 + 1
    S = np.stack([_spec(rng, T, F), _spec(rng, T, F, real=True)])
    out = p.batch_lws(S)
    name = p.plan().last_kernel()["name"]
    assert name.startswith("systolic_fp64_q"), name
    assert name.endswith("_wide") == (1200 <= fsize <= 2100), name       # 128 frames in flight from ~525 bins on (where 64 lanes hold one slot only)
    assert name.endswith("_xwide") == (fsize > 2100), name
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    for b in range(2):
        ref = oracle.batch_lws(S[b], p.W, thr)
        err = np.abs(out[b] - ref).max() / np.abs(ref).max()
        print("lws(%d,%d) T=%d iters=%d %s input: max err / max value = %.2e" % (fsize, fshift, T, iters, "real" if b else "complex", err))
        assert err < (1e-10 if b else 1e-11), (b, err)


def test_same_as_generic_engine_to_rounding_and_thresholds_skip_bins():
    """the default schedule (thresholds that decay: most bins are skipped in the first sweeps) on 100 sweeps, both engines.
    The two fp64 engines agree to 1e-10 on random and on zero phases (observed 5e-14 / 3e-14; before the DC / Nyquist bins were
    kept exactly real the zero-phase run differed by 2.6e-8 at this size and by O(1) on 5 % of the bins at 500 frames)."""
    rng = np.random.default_rng(5)
    T, F = 150, 513
    mag = np.abs(_spec(rng, T, F)) * rng.random((T, F)) ** 4      # wide dynamic range: the thresholds matter
    p = lws_amd.lws(1024, 256, precision="fp64")
    q = lws_amd.lws(1024, 256, precision="fp64", force_generic=True)
    for name, S, bar in (("random phases", mag * np.exp(2j * np.pi * rng.random((T, F))), 1e-10), ("zero phases", mag.astype(complex), 1e-10)):
        a = p.batch_lws(S)
        assert p.plan().last_kernel()["name"] == "systolic_fp64_q4"
        b = q.batch_lws(S)
        assert q.plan().last_kernel()["name"].startswith("generic")
        err = np.abs(a - b).max() / np.abs(b).max()
        print("100 sweeps, default schedule, %s, fp64 systolic vs generic: %.2e" % (name, err))
        assert err < bar, (name, err)
        assert np.abs(np.abs(a) - np.abs(S)).max() < 1e-12 * np.abs(S).max()


def test_config2_spectrogram_against_the_oracle(oracle):
    """one spectrogram of BASELINE config 2's shape (500 x 513, lws(1024,256)), the reference's default schedule of 100 sweeps
    (the oracle needs ~1 s): random phases and zero phases -- the documented usage run_lws(np.abs(X)) -- to 1e-10 (observed 2e-12 / 8e-13)"""
    rng = np.random.default_rng(2)
    M = np.abs(rng.standard_normal((500, 513)) + 1j * rng.standard_normal((500, 513)))
    p = lws_amd.lws(1024, 256, precision="fp64")
    thr = lws_amd.get_thresholds(100, 100, 0.1, 1)
    for name, S, bar in (("random phases", M * np.exp(2j * np.pi * rng.random(M.shape)), 1e-10), ("zero phases", M.astype(complex), 1e-10)):
        out = p.batch_lws(S)
        k = p.plan().last_kernel()
        assert k["name"] == "systolic_fp64_q4" and k["launches"] == 25
        ref = oracle.batch_lws(S, p.W, thr)
        err = np.abs(out - ref).max() / np.abs(ref).max()
        print("config-2 spectrogram, default schedule, %s: max err / max value = %.2e" % (name, err))
        assert err < bar, (name, err)
        assert np.abs(np.abs(out) - M).max() < 1e-12 * M.max()


@pytest.mark.parametrize("fsize,fshift", [(1024, 256), (1016, 254), (1020, 510), (400, 100)])
def test_zero_phase_start_stays_on_the_reference_trajectory(fsize, fshift, oracle):
    """real-valued input, the default schedule's 100 sweeps: the DC and Nyquist bins stay exactly real as in the reference (half-lengths
    0 and 4 mod 8, and 2 mod 4 with Q = 2: weight row 0) -- else 5 % of the bins end up O(1) away"""
    rng = np.random.default_rng(fsize)
    T, F = 140, fsize // 2 + 1
    M = np.abs(rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F)))
    p = lws_amd.lws(fsize, fshift, precision="fp64")
    out = p.batch_lws(M)
    assert p.plan().last_kernel()["name"].startswith("systolic_fp64_q")
    ref = oracle.batch_lws(M.astype(complex), p.W, lws_amd.get_thresholds(100, 100, 0.1, 1))
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print("lws(%d,%d) zero phases, 100 sweeps: max err / max value = %.2e" % (fsize, fshift, err))
    assert err < 1e-10, err
    assert np.all(out[:, 0].imag == 0) and np.all(out[:, -1].imag == 0) and np.all(ref[:, -1].imag == 0)


def test_zero_phase_start_where_the_reference_is_its_own_noise(oracle):
    """lws(996,249): Q = 4 and a half-length that is 2 mod 4 put the Nyquist bin on weight row 2, whose k = 0 weights are
    base x exp(j pi r) -- with an imaginary part of 1e-16 from numpy's exp.  That seeds the unstable line: the REFERENCE's Nyquist
    bins leave the real axis (asserted), and its zero-phase trajectory is that noise amplified.  The kernel uses the exact (-1)^r and
    keeps the Nyquist bin of a real signal's spectrogram real; the two agree on the median bin and in quality, not bin by bin."""
    rng = np.random.default_rng(996)
    T, F = 140, 499
    M = np.abs(rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F)))
    p = lws_amd.lws(996, 249, precision="fp64")
    out = p.batch_lws(M)
    assert p.plan().last_kernel()["name"] == "systolic_fp64_q4"
    ref = oracle.batch_lws(M.astype(complex), p.W, lws_amd.get_thresholds(100, 100, 0.1, 1))
    assert np.abs(ref[:, -1].imag).max() > 1e-3 * np.abs(ref).max()          # the reference's Nyquist bins are far from real
    assert np.all(out[:, 0].imag == 0) and np.all(out[:, -1].imag == 0)
    d = np.abs(out - ref)
    assert np.median(d) < 1e-9 * np.abs(ref).max()
    assert abs(p.get_consistency(out) - p.get_consistency(ref)) < 0.05
    assert np.abs(np.abs(out) - M).max() < 1e-12 * M.max()
    # 30 sweeps in, before the noise has grown: bin by bin
    thr = lws_amd.get_thresholds(100, 100, 0.1, 1)[:45]
    o45 = lws_amd.lws(996, 249, precision="fp64", batch_iterations=45).batch_lws(M, thresholds=thr)
    r45 = oracle.batch_lws(M.astype(complex), p.W, thr)
    assert np.abs(o45 - r45).max() < 1e-9 * np.abs(r45).max()


@pytest.mark.parametrize("wname", ["hamming", "blackman"])
def test_other_windows_run_the_build_with_every_tap(wname, oracle):
    """the default (sqrt-Hann) windows leave 6 of the 23 weights of a row at zero and get a build without those taps; any other
    window -- here Hamming / Blackman analysis windows, whose tensors have all 23 -- runs the full build"""
    rng = np.random.default_rng(13)
    win = getattr(np, wname)(256)
    for fshift, T in ((64, 70), (128, 40)):
        p = lws_amd.lws(win, fshift, batch_iterations=8, batch_alpha=1.0, precision="fp64")
        assert np.count_nonzero(np.abs(p.W[0]) > 1e-12) > (17 if p.W.shape[1] == 4 else 7)
        S = np.stack([_spec(rng, T, 129), _spec(rng, T, 129, real=True)])
        out = p.batch_lws(S)
        assert p.plan().last_kernel()["name"].startswith("systolic_fp64_q"), p.plan().last_kernel()["name"]
        thr = lws_amd.get_thresholds(8, 1.0, 0.1, 1)
        for b in range(2):
            ref = oracle.batch_lws(S[b], p.W, thr)
            err = np.abs(out[b] - ref).max() / np.abs(ref).max()
            assert err < 1e-10, (wname, fshift, b, err)


def test_general_weight_tensor_with_row_per_bin(oracle):
    """use_simplifications=False (lws.pyx:246-253: LWSfractionalQ on a tensor with one row per bin, Qp = N): the rows are still the
    quarter turns of row 0, so the fp64 systolic engine takes it"""
    rng = np.random.default_rng(17)
    p = lws_amd.lws(256, 64)
    Wg = lws_amd.create_weights(p.awin, p.swin, 64, 5, use_summarized_weights=False)
    assert Wg.shape[0] > 4
    S = _spec(rng, 50, 129)
    thr = lws_amd.get_thresholds(6, 1.0, 0.1, 1)
    out = lws_amd.batch_lws(S, Wg, thr, use_simplifications=False, precision="fp64")
    ref = oracle.batch_lws(S, Wg, thr)
    assert np.abs(out - ref).max() < 1e-11 * np.abs(ref).max()
    plan = _capi.Plan(129, Wg, precision="fp64")
    plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "systolic_fp64_q4"
    plan.close()


def test_device_resident_and_repeatable():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    S = _spec(rng, 40, 513)
    p = lws_amd.lws(1024, 256, batch_iterations=10, precision="fp64")
    a = p.batch_lws(S)
    b = p.batch_lws(S)
    assert np.array_equal(a, b)            # the schedule is fixed: run to run bit-identical
    plan = _capi.Plan(513, p.W, precision="fp64")
    d = torch.from_numpy(np.stack([S, S])).cuda()
    plan.batch_dev(d.data_ptr(), 2, 40, lws_amd.get_thresholds(10, p.batch_alpha, p.batch_beta, p.batch_gamma))
    torch.cuda.synchronize()
    assert plan.last_kernel()["name"] == "systolic_fp64_q4" and plan.last_kernel()["launches"] == 3   # four sweep slots per pass
    out = d.cpu().numpy()
    assert np.array_equal(out[0], a) and np.array_equal(out[1], a)
    plan.close()


def test_two_short_spectrograms_share_a_workgroup(oracle):
    """frames of up to ~300 bins: two spectrograms side by side in a wave (32 lanes each); an odd batch leaves half a workgroup idle"""
    rng = np.random.default_rng(21)
    for fsize, fshift, T in ((512, 128, 45), (400, 100, 70), (128, 64, 33)):
        F = fsize // 2 + 1
        p = lws_amd.lws(fsize, fshift, batch_iterations=6, batch_alpha=1.0, precision="fp64")
        S = np.stack([_spec(rng, T, F) * (b + 1) for b in range(3)])       # different mean |S|: each spectrogram its own thresholds
        out = p.batch_lws(S)
        assert p.plan().last_kernel()["name"].startswith("systolic_fp64_q")
        thr = lws_amd.get_thresholds(6, 1.0, 0.1, 1)
        for b in range(3):
            ref = oracle.batch_lws(S[b], p.W, thr)
            assert np.abs(out[b] - ref).max() < 1e-11 * np.abs(ref).max(), (fsize, b)


def test_large_batches_go_through_the_scratch_in_chunks(monkeypatch):
    """LWS_S64_CHUNK=2: five spectrograms as 2 + 2 + 1, same bits as in one piece"""
    rng = np.random.default_rng(8)
    for fsize, fshift, T in ((1024, 256, 12), (256, 64, 20)):
        F = fsize // 2 + 1
        S = np.stack([_spec(rng, T, F) * (1 + b) for b in range(5)])
        p = lws_amd.lws(fsize, fshift, batch_iterations=5, batch_alpha=1.0, precision="fp64")
        whole = p.batch_lws(S)
        monkeypatch.setenv("LWS_S64_CHUNK", "2")
        lws_amd.clear_plan_cache()
        q = lws_amd.lws(fsize, fshift, batch_iterations=5, batch_alpha=1.0, precision="fp64")
        parts = q.batch_lws(S)
        assert q.plan().last_kernel()["name"].startswith("systolic_fp64") and q.plan().last_kernel()["launches"] == 3 * (2 if F > 300 else 2)
        monkeypatch.delenv("LWS_S64_CHUNK")
        lws_amd.clear_plan_cache()
        assert np.array_equal(whole, parts)


@pytest.mark.parametrize("fsize,fshift,T,iters", [(2048, 512, 700, 3), (2048, 1024, 530, 5), (1536, 384, 400, 4)])
def test_wide_geometry_over_many_blocks_of_frames(fsize, fshift, T, iters):
    """128 frames in flight: four to six blocks of 128 frames, complex and magnitudes-only input, against the order-exact engine
    (to rounding: the sums are taken in another order)."""
    rng = np.random.default_rng(T)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, batch_iterations=iters, batch_alpha=1.0, precision="fp64")
    pg = lws_amd.lws(fsize, fshift, batch_iterations=iters, batch_alpha=1.0, precision="fp64", force_generic=True)
    S = np.stack([_spec(rng, T, F), _spec(rng, T, F, real=True)])
    out = p.batch_lws(S)
    assert p.plan().last_kernel()["name"].endswith("_wide")
    ref = pg.batch_lws(S)
    assert pg.plan().last_kernel()["name"] == "generic_skew_fp64"
    assert np.abs(out - ref).max() < 1e-10 * np.abs(ref).max()


def test_unsupported_shapes_fall_back():
    rng = np.random.default_rng(3)
    # 2101 bins, Q = 3, Q = 8: the band engine in fp64 (round 6; the generic engine before); 4097 bins: no fp64 ring of that frame fits
    for fsize, fshift, want in ((8192, 2048, "generic"), (4200, 1050, "band_fp64"), (60, 20, "band_fp64"), (64, 8, "band_fp64")):
        p = lws_amd.lws(fsize, fshift, batch_iterations=2, precision="fp64")
        p.batch_lws(_spec(rng, 6, fsize // 2 + 1))
        assert p.plan().last_kernel()["name"].startswith(want), (fsize, fshift, p.plan().last_kernel())


def test_random_shapes_against_the_oracle():
    """tools/stress_sys64.py: frame sizes around the period / ring-size boundaries, frame counts around multiples of 64, sweep counts
    that are no multiple of the slots per pass, thresholds that skip bins -- 60 cases, each <= 1e-10 of the largest value."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_sys64.py"), "60", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst" in r.stdout
