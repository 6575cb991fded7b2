"""GPU (-m gpu): the band engine (lws_amd/csrc/lws_band.hip) -- batch sweeps of the shapes no systolic build takes -- against the
oracle (LWSanyQ / LWSfractionalQ, lwslib.cpp:283-467) at SURVEY 8c's bars in fp32 and to rounding in fp64, against the order-exact
generic engine, and against itself under every geometry the launcher could choose (the geometry is scheduling only: same bits)."""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def spectrograms(B, T, F, seed, scale=None):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    if scale is not None:
        S *= np.asarray(scale, dtype=float)[:, None, None]
    return S


def check_fp32(out, ref, S):
    mean = np.mean(np.abs(S))
    d = np.abs(out - ref)
    assert rel_l2(out, ref) < 1e-3, rel_l2(out, ref)
    assert np.median(d) < 1e-6 * mean, np.median(d) / mean
    assert np.quantile(d, 0.999) < 1e-3 * mean
    assert np.abs(np.abs(out) - np.abs(S)).max() < 1e-6 * np.abs(S).max()      # magnitudes are the targets'


# the verdict's three shapes first; then every family the engine serves
SHAPES = [
    # fsize, fshift, L, T, sweeps
    (2048, 256, 5, 70, 5),       # Q = 8 above 513 bins (87.5 % overlap at config 5's frame size)
    (1024, 64, 5, 40, 4),        # sixteen frames per stencil row
    (1024, 256, 8, 70, 6),       # stencil of half-width 8
    (1024, 256, 10, 40, 3), (2048, 512, 6, 40, 3), (2048, 512, 7, 40, 3), (1024, 128, 7, 30, 3),
    (1200, 100, 5, 40, 3),       # Q = 12
    (2000, 400, 5, 40, 3), (2004, 334, 5, 40, 3), (2044, 292, 5, 30, 3),     # Q = 5, 6, 7 above 513 bins
    (2048, 320, 5, 30, 3),       # fractional Q above 4 on long frames (general tensor, LWSfractionalQ)
    (8192, 2048, 5, 20, 3),      # 4097 bins
    (44, 11, 5, 9, 4), (36, 9, 5, 30, 3),     # frames the narrow build refuses (F - 1 below 24 with a frame end inside a block)
]


@pytest.mark.parametrize("fsize,fshift,L,T,sweeps", SHAPES)
def test_band_engine_against_the_oracle_fp32(oracle, fsize, fshift, L, T, sweeps):
    p = lws_amd.lws(fsize, fshift, L=L)
    F = fsize // 2 + 1
    S = spectrograms(2, T, F, seed=fsize + T, scale=[1.0, 40.0])
    thr = lws_amd.get_thresholds(sweeps, 1.0, 0.3, 1)
    plan = _capi.Plan(F, p.W)
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp32", plan.last_kernel()
    for b in range(2):
        check_fp32(out[b], oracle.batch_lws(S[b], p.W, thr), S[b])
    # ... and what the order-exact fp32 engine gives: the two differ by rounding only
    gen = _capi.Plan(F, p.W, force_generic=True)
    ref32 = gen.batch(S, thr)
    assert gen.last_kernel()["name"].startswith("generic")
    assert rel_l2(out, ref32) < 1e-3
    plan.close(); gen.close()


@pytest.mark.parametrize("fsize,fshift,L,T,sweeps", [(1024, 128, 5, 40, 5), (2048, 256, 5, 20, 3), (1024, 64, 5, 20, 3), (1024, 256, 8, 40, 4),
                                                      (768, 256, 5, 40, 4), (1000, 200, 5, 40, 4), (1024, 256, 3, 40, 4), (4200, 1050, 5, 20, 3)])
def test_band_engine_fp64_is_the_reference_to_rounding(oracle, fsize, fshift, L, T, sweeps):
    """What the fp64 systolic engine does not take -- Q other than 2 and 4, table twiddles, general tensors, other stencil widths --
    on an fp64 plan: the reference's arithmetic type, a bin's sum in another order."""
    p = lws_amd.lws(fsize, fshift, L=L)
    F = fsize // 2 + 1
    S = spectrograms(2, T, F, seed=7 * fsize + T, scale=[1.0, 1e-3])
    thr = lws_amd.get_thresholds(sweeps, 1.0, 0.3, 1)
    plan = _capi.Plan(F, p.W, precision="fp64")
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp64", plan.last_kernel()
    for b in range(2):
        ref = oracle.batch_lws(S[b], p.W, thr)
        assert np.abs(out[b] - ref).max() < 1e-10 * np.abs(ref).max(), np.abs(out[b] - ref).max() / np.abs(ref).max()
    plan.close()


def test_fp64_plans_with_general_tensors_stay_on_the_order_exact_engine():
    """The rows of the tensors create_weights builds for a hop that does not divide the frame (lws.pyx:164-181) are twiddle images of
    row 0 to 1e-13 only (numpy's exp of an angle of up to N turns); the band engine works from row 0 and exact twiddles, which in fp32
    is below rounding and in fp64 would be 1e-9 after a few sweeps -- not the reference's values: such fp64 plans are refused."""
    p = lws_amd.lws(1024, 384)
    S = spectrograms(1, 20, 513, seed=2)[0]
    plan = _capi.Plan(513, p.W, precision="fp64")
    plan.batch(S, [0.0, 0.0])
    assert plan.last_kernel()["name"].startswith("generic"), plan.last_kernel()
    plan.close()


def test_the_geometry_is_scheduling_only(monkeypatch):
    """Steps between frames, frames in flight, sweep slots per pass and the cut of the batch into chunks change when a bin is
    computed, never what it is computed from or in which order: identical bits."""
    p = lws_amd.lws(1024, 128)
    S = spectrograms(3, 150, 513, seed=3, scale=[1.0, 7.0, 0.2])
    thr = lws_amd.get_thresholds(7, 1.0, 0.2, 1)
    plan = _capi.Plan(513, p.W)
    sys_out = plan.batch(S, thr)
    assert plan.last_kernel()["name"].startswith("systolic")
    # (this plan normally runs on the systolic Q = 8 build: a plan created with the systolic builds switched off takes the band engine)
    monkeypatch.setenv("LWS_NO_SYSTOLIC", "1")
    plan2 = _capi.Plan(513, p.W)
    monkeypatch.delenv("LWS_NO_SYSTOLIC")
    ref = plan2.batch(S, thr)
    assert plan2.last_kernel()["name"] == "band_fp32", plan2.last_kernel()
    assert rel_l2(ref, sys_out) < 1e-3
    for env in (dict(LWS_BAND_SKW="7"), dict(LWS_BAND_SKW="11"), dict(LWS_BAND_NLS="128"), dict(LWS_BAND_NS="1"), dict(LWS_BAND_NS="3"),
                dict(LWS_BAND_CHUNK="2"), dict(LWS_BAND_CHUNK="1", LWS_BAND_NLS="256")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(plan2.batch(S, thr), ref), env
        for k in env:
            monkeypatch.delenv(k)
    plan.close(); plan2.close()


def test_the_geometry_is_scheduling_only_l8(monkeypatch):
    p = lws_amd.lws(1024, 256, L=8)
    S = spectrograms(3, 150, 513, seed=4, scale=[1.0, 7.0, 0.2])
    thr = lws_amd.get_thresholds(9, 1.0, 0.2, 1)
    plan = _capi.Plan(513, p.W)
    ref = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp32"
    for env in (dict(LWS_BAND_SKW="12"), dict(LWS_BAND_SKW="17"), dict(LWS_BAND_NLS="128"), dict(LWS_BAND_NS="1"), dict(LWS_BAND_NS="3"),
                dict(LWS_BAND_CHUNK="2"), dict(LWS_BAND_CHUNK="1", LWS_BAND_NLS="256")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(plan.batch(S, thr), ref), env
        for k in env:
            monkeypatch.delenv(k)
    plan.close()


def test_helper_waves_change_rounding_only(monkeypatch):
    """The exact builds share a slot's frame offsets between a main wave and helper waves (LWS_BAND_NO_HELPERS=1: one wave per slot):
    a bin's sum is then a sum of partial sums -- the same values to fp32 rounding, not the same bits."""
    for fsize, fshift, L in ((2048, 256, 5), (1024, 64, 5), (1024, 256, 8)):
        p = lws_amd.lws(fsize, fshift, L=L)
        F = fsize // 2 + 1
        S = spectrograms(2, 140, F, seed=fsize + L)
        thr = lws_amd.get_thresholds(5, 1.0, 0.2, 1)
        plan = _capi.Plan(F, p.W)
        with_h = plan.batch(S, thr)
        monkeypatch.setenv("LWS_BAND_NO_HELPERS", "1")
        without = plan.batch(S, thr)
        monkeypatch.delenv("LWS_BAND_NO_HELPERS")
        assert plan.last_kernel()["name"] == "band_fp32"
        assert not np.array_equal(with_h, without) and rel_l2(with_h, without) < 1e-4, (fsize, fshift, L, rel_l2(with_h, without))
        plan.close()


def test_passes_without_an_active_bin_are_dropped_per_spectrogram():
    """The reference's default schedule starts with ~38 sweeps no bin takes part in; the band engine skips a pass of its
    sweep slots for a spectrogram whose largest magnitude is below all of the pass's thresholds -- per spectrogram: one of another scale in
    the same batch has another set of such passes.  Results equal the oracle's either way."""
    from oracle.oracle import Oracle
    orc = Oracle()
    p = lws_amd.lws(2048, 256)
    S = np.abs(spectrograms(2, 30, 1025, seed=5, scale=[1.0, 300.0])).astype(np.complex128)
    thr = np.array([50.0, 40.0, 3.9, 3.8, 1.0, 0.9, 1e9, 1e9, 0.5])      # x mean|S| of each spectrogram: the largest Rayleigh magnitude is ~3.8-4.5 x the mean
    plan = _capi.Plan(1025, p.W)
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp32"
    for b in range(2):
        check_fp32(out[b], orc.batch_lws(S[b], p.W, thr), S[b])
    plan.close()


def test_dropped_sweeps_and_untouched_bins():
    """Sweeps whose threshold no bin exceeds change nothing; a bin no sweep updates comes back bit for bit (complex128)."""
    p = lws_amd.lws(2048, 256)
    S = spectrograms(2, 40, 1025, seed=11)
    thr = np.array([1e9, 3.0, 1e9, 2.5])          # of mean |S|: only the largest bins take part
    plan = _capi.Plan(1025, p.W)
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp32"
    untouched = np.abs(S) <= 2.5 * np.mean(np.abs(S), axis=(1, 2), keepdims=True)
    assert untouched.mean() > 0.9 and np.array_equal(out[untouched], S[untouched])
    assert not np.array_equal(out[~untouched], S[~untouched])
    plan.close()


def test_zero_phase_start_keeps_dc_and_nyquist_real(oracle):
    """Magnitudes-only input (python/README.md:98-100): the DC and Nyquist lines stay exactly real, as in the reference."""
    p = lws_amd.lws(2048, 256)
    S = np.abs(spectrograms(1, 60, 1025, seed=5))[0].astype(np.complex128)
    thr = np.zeros(6)
    plan = _capi.Plan(1025, p.W, precision="fp64")
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp64"
    ref = oracle.batch_lws(S, p.W, thr)
    assert np.all(out[:, 0].imag == 0) and np.all(out[:, -1].imag == 0)
    assert np.all(ref[:, 0].imag == 0) and np.all(ref[:, -1].imag == 0)
    assert np.abs(out - ref).max() < 1e-9 * np.abs(ref).max()
    plan.close()


def test_long_run_at_full_width(oracle):
    """lws(2048,256), 300 frames (three blocks of 128 lanes), 30 dense sweeps from random phases: value level against the oracle."""
    p = lws_amd.lws(2048, 256)
    S = spectrograms(1, 300, 1025, seed=21)[0]
    thr = np.zeros(30)
    plan = _capi.Plan(1025, p.W)
    out = plan.batch(S, thr)
    assert plan.last_kernel()["name"] == "band_fp32"
    check_fp32(out, oracle.batch_lws(S, p.W, thr), S)
    plan.close()


def test_run_lws_of_a_plan_whose_batch_stage_is_the_band_engine():
    """lws.lws(2048, 256).run_lws: the README call on a plan with 87.5 % overlap -- no stage on the generic engine."""
    import warnings
    p = lws_amd.lws(2048, 256, mode="music", batch_iterations=20)
    S = np.abs(spectrograms(1, 50, 1025, seed=9))[0]
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # a plan that falls back to the generic engine warns
        out = p.run_lws(S)
    assert p.plan().last_kernel()["name"] == "band_fp32"
    assert np.abs(np.abs(out) - S).max() < 1e-5 * S.max()
