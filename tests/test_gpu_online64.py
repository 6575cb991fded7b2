"""GPU (-m gpu): the online stage of an fp64 plan on its LDS engine (lws_online64.hip): every sum in the order-exact generic engine's
order, so the results must equal generic_fp64's BIT FOR BIT, and the reference goldens to 1e-8 (TF_RTISI_LA, lwslib.cpp:1424-1492)."""
import numpy as np
import pytest

import lws_amd
from conftest import load_golden
from lws_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def q8_on_this_engine(monkeypatch):
    """fp64 plans of Q = 8 run their online stage on the team engine's order-exact kernel by default (1.3x faster, the same bits: the
    generic engine's -- tests/test_gpu_team.py); this module is about lws_online64.hip, whose Q = 8 kernel LWS_NO_TEAM_Q8=1 selects."""
    monkeypatch.setenv("LWS_NO_TEAM_Q8", "1")


def weights(tag):
    h = load_golden("helpers.npz")
    return h[f"W_{tag}"], h[f"W_ai_{tag}"], h[f"W_af_{tag}"]


@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8", "48_16"])
def test_reference_goldens_and_generic_bits(tag):
    g = load_golden("wrappers.npz")
    W, W_ai, W_af = weights(tag)
    S, thr = g[f"S_{tag}"], g[f"thr_{tag}"]
    F = S.shape[1]
    qdiv = 2 * (F - 1) / int(tag.split("_")[1])
    lds = _capi.Plan(F, W, W_ai, W_af, precision="fp64")
    gen = _capi.Plan(F, W, W_ai, W_af, precision="fp64", force_generic=True)
    for thr_n, LA, key in ((3, 3, "online"), (3, 0, "online_la0"), (2, 5, "online_la5")):
        out = lds.online(S, thr[:thr_n], LA, qdiv)
        assert lds.last_kernel()["name"] == "online_lds_fp64"
        assert np.abs(out - g[f"{key}_{tag}"]).max() < 1e-8
        ref = gen.online(S, thr[:thr_n], LA, qdiv)
        assert gen.last_kernel()["name"] == "generic_fp64"
        assert np.array_equal(out, ref)
    lds.close(); gen.close()


@pytest.mark.parametrize("fsize,fshift,T,LA,iters", [(1024, 256, 40, 3, 10), (1024, 512, 30, 3, 4), (1024, 128, 24, 3, 3), (512, 128, 70, 5, 6),
                                                     (768, 256, 30, 2, 5), (64, 16, 200, 3, 10), (1000, 250, 25, 0, 4), (1024, 256, 9, 7, 2), (1024, 256, 3, 3, 3)])
def test_bit_identical_to_the_generic_engine(fsize, fshift, T, LA, iters):
    """Config-3-like shapes (and Q = 2 / 8 / 3, short and long look-aheads, fewer frames than the look-ahead): complex input and
    magnitudes-only input, stacks of spectrograms of different scale."""
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", look_ahead=LA, online_iterations=iters)
    pg = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", look_ahead=LA, online_iterations=iters, force_generic=True)
    S = rng.standard_normal((3, T, F)) + 1j * rng.standard_normal((3, T, F))
    S[1] = np.abs(S[1])
    S[2] *= 1e-3
    out = p.online_lws(S)
    assert p.plan().last_kernel()["name"] == "online_lds_fp64"
    ref = pg.online_lws(S)
    assert pg.plan().last_kernel()["name"] == "generic_fp64"
    assert np.array_equal(out, ref)
    assert np.isfinite(out).all() and np.abs(np.abs(out) - np.abs(S)).max() < 1e-9 * np.abs(S).max()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("fsize,fshift,T,LA,iters", [(1024, 256, 24, 3, 10), (1024, 512, 20, 3, 4), (768, 256, 18, 2, 5), (64, 16, 90, 5, 6), (2048, 512, 10, 3, 10),
                                                     (1024, 128, 14, 3, 4), (64, 8, 60, 2, 5)])
def test_two_waves_per_spectrogram_do_not_race(fsize, fshift, T, LA, iters, seed, monkeypatch):
    """k_online64p: the even bin of a step on one wave, the odd bin on the other, the neighbour frames' products formed up to one and a half
    steps ahead.  What a half-step reads must not depend on how far the other wave has got inside it: with the waves idling for
    pseudo-random times inside their half-steps (LWS_ONLINE64_STRESS) the results are the same bits -- the one-wave kernel's and the
    generic engine's."""
    rng = np.random.default_rng(fsize + T + seed)
    F = fsize // 2 + 1
    kw = dict(mode="music", precision="fp64", look_ahead=LA, online_iterations=iters)
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    S[1] = np.abs(S[1])
    ref = lws_amd.lws(fsize, fshift, force_generic=True, **kw).online_lws(S)
    monkeypatch.setenv("LWS_ONLINE64_STRESS", str(seed))
    p = lws_amd.lws(fsize, fshift, **kw)
    out = p.online_lws(S)
    assert p.plan().last_kernel()["name"] == "online_lds_fp64"
    assert np.array_equal(out, ref)
    monkeypatch.delenv("LWS_ONLINE64_STRESS")
    monkeypatch.setenv("LWS_ONLINE64_ONE_WAVE", "1")      # (read on every launch)
    p1 = lws_amd.lws(fsize, fshift, **kw)
    assert np.array_equal(p1.online_lws(S), ref)
    assert p1.plan().last_kernel()["name"] == "online_lds_fp64_1w"      # the one-wave kernel did run


def test_frames_too_long_for_fp64_rows_keep_the_generic_engines_bits():
    p = lws_amd.lws(4096, 1024, mode="music", precision="fp64", online_iterations=2)
    S = np.abs(np.random.default_rng(0).standard_normal((6, 2049))).astype(complex)
    p.online_lws(S)
    # (no fp64 ring holds 4096-point frames, and the 348 bins of a wavefront step x 39 increments do not fit the team engine's order-exact
    #  kernel either: the generic engine)
    assert p.plan().last_kernel()["name"] == "generic_fp64"


@pytest.mark.parametrize("fsize,fshift,T,LA,iters", [(2048, 512, 14, 3, 3), (2048, 1024, 9, 3, 10), (2048, 512, 30, 0, 2), (1536, 384, 20, 5, 4), (2048, 512, 5, 4, 2)])
def test_2048_point_frames_keep_their_magnitudes_in_memory(fsize, fshift, T, LA, iters):
    """Frames whose window of state rows AND magnitudes does not fit the LDS (from ~700 bins): k_online64<Q, false> keeps the state rows
    only and reads a bin's target magnitude from memory while its taps are summed -- the same sums, the same bits."""
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", look_ahead=LA, online_iterations=iters)
    pg = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", look_ahead=LA, online_iterations=iters, force_generic=True)
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    S[1] = np.abs(S[1]) * 1e2
    out = p.online_lws(S)
    assert p.plan().last_kernel()["name"] == "online_lds_fp64"
    assert np.array_equal(out, pg.online_lws(S))
    assert pg.plan().last_kernel()["name"] == "generic_fp64"
    assert np.isfinite(out).all() and np.abs(np.abs(out) - np.abs(S)).max() < 1e-9 * np.abs(S).max()


def test_run_lws_music_fp64_uses_it():
    rng = np.random.default_rng(5)
    M = np.abs(rng.standard_normal((40, 513)) + 1j * rng.standard_normal((40, 513)))
    p = lws_amd.lws(1024, 256, mode="music", precision="fp64", batch_iterations=20)
    pg = lws_amd.lws(1024, 256, mode="music", precision="fp64", batch_iterations=20, force_generic=True)
    s1 = p.online_lws(p.nofuture_lws(M))
    assert p.plan().last_kernel()["name"] == "online_lds_fp64"
    assert np.array_equal(s1, pg.online_lws(pg.nofuture_lws(M)))
    out = p.run_lws(M)
    assert np.abs(np.abs(out) - M).max() < 1e-12 * M.max()


def test_run_lws_music_fp64_of_a_2048_point_plan_has_no_generic_stage():
    rng = np.random.default_rng(6)
    M = np.abs(rng.standard_normal((2, 30, 1025)) + 1j * rng.standard_normal((2, 30, 1025)))
    p = lws_amd.lws(2048, 512, mode="music", precision="fp64", batch_iterations=12)
    pg = lws_amd.lws(2048, 512, mode="music", precision="fp64", batch_iterations=12, force_generic=True)
    s0 = p.nofuture_lws(M)
    assert p.plan().last_kernel()["name"].startswith("nofuture_lds") and p.plan().last_kernel()["name"].endswith("_fp64")
    assert np.array_equal(s0, pg.nofuture_lws(M))
    s1 = p.online_lws(s0)
    assert p.plan().last_kernel()["name"] == "online_lds_fp64"
    assert np.array_equal(s1, pg.online_lws(s0))
    s2 = p.batch_lws(s1)
    assert p.plan().last_kernel()["name"] == "systolic_fp64_q4_wide"
    ref = pg.batch_lws(s1)
    assert np.abs(s2 - ref).max() < 1e-10 * np.abs(ref).max()
    out = p.run_lws(M)
    assert np.abs(np.abs(out) - M).max() < 1e-12 * M.max()


# ---------------------------------------------------------------------------------------------- no-future sweeps of an fp64 plan
@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8", "48_16"])
def test_nofuture_goldens_and_generic_bits(tag):
    """lws_nofuture.hip's one-lane-per-bin variant in double (round 5): NoFuture_LWSQ2 / anyQ and the shipped NoFuture_LWSQ4 addressing
    (lwslib.cpp:473-690) -- the reference goldens to 1e-8 and generic_fp64's results bit for bit."""
    g = load_golden("wrappers.npz")
    W, W_ai, W_af = weights(tag)
    S, thr = g[f"S_{tag}"], g[f"thr_{tag}"]
    F = S.shape[1]
    lds = _capi.Plan(F, W, W_ai, W_af, precision="fp64")
    gen = _capi.Plan(F, W, W_ai, W_af, precision="fp64", force_generic=True)
    out = lds.nofuture(S, thr[:2], wsel=_capi.LWS_W_AI)
    assert lds.last_kernel()["name"] in ("nofuture_lds_fp64", "nofuture_lds_q4compat_fp64")
    assert np.abs(out - g[f"nofuture_{tag}"]).max() < 1e-8
    assert np.array_equal(out, gen.nofuture(S, thr[:2], wsel=_capi.LWS_W_AI)) and gen.last_kernel()["name"] == "generic_fp64"
    out = lds.nofuture(np.abs(S), thr, wsel=_capi.LWS_W)
    assert np.array_equal(out, gen.nofuture(np.abs(S), thr, wsel=_capi.LWS_W))
    lds.close(); gen.close()


@pytest.mark.parametrize("fsize,fshift,T", [(1024, 256, 60), (1024, 512, 40), (1024, 128, 30), (512, 128, 90), (768, 256, 40), (2048, 512, 12), (1000, 250, 30)])
def test_nofuture_bit_identical_to_the_generic_engine(fsize, fshift, T):
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", nofuture_iterations=2)
    pg = lws_amd.lws(fsize, fshift, mode="music", precision="fp64", nofuture_iterations=2, force_generic=True)
    S = rng.standard_normal((3, T, F)) + 1j * rng.standard_normal((3, T, F))
    S[1] = np.abs(S[1])
    S[2] *= 1e3
    out = p.nofuture_lws(S)
    assert p.plan().last_kernel()["name"].startswith("nofuture_lds") and p.plan().last_kernel()["name"].endswith("_fp64")
    assert np.array_equal(out, pg.nofuture_lws(S)) and pg.plan().last_kernel()["name"] == "generic_fp64"


def test_general_weights_of_an_fp64_plan_stay_on_the_generic_engine():
    """The rows of a general tensor (use_simplifications=False) repeat to 1e-9, not to the bit: an fp64 plan keeps the order-exact engine."""
    p = lws_amd.lws(64, 16, mode="music", precision="fp64", use_simplifications=False, nofuture_iterations=1)
    S = np.abs(np.random.default_rng(0).standard_normal((9, 33))).astype(complex)
    p.nofuture_lws(S)
    assert p.plan().last_kernel()["name"] == "generic_fp64"
