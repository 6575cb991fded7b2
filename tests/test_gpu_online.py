"""The LDS-resident online engine (lws_amd/csrc/lws_online.hip) behind lws_online_lws / online_lws().

Two anchors.  (1) At small sizes, where fp32 rounding stays rounding, it is checked against the fp64 oracle like every
other fp32 path.  (2) At realistic sizes the online algorithm amplifies rounding differences (two correct fp32
engines end up as far from each other as from fp64), so there the engine's schedule, frame window and slot logic are
pinned bit for bit: its verification variant (LWS_ONLINE_SERIAL_TAPS=1: same kernel, taps summed by one lane in the
generic engine's order) must reproduce the order-exact generic engine's fp32 result exactly.
"""
import os

import numpy as np
import pytest

import lws_amd
from lws_amd import _capi
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _online(F, W, S, thr, LA, qdiv, **kw):
    plan = _capi.Plan(F, W[0], W[1], W[2], **kw)
    out = plan.online(S, thr, LA, qdiv)
    name = plan.last_kernel()["name"]
    plan.close()
    return out, name


@pytest.mark.parametrize("fsize,fshift,T,LA,iters,B", [
    (512, 128, 100, 3, 10, 2),     # Q = 4, config-1 frame size
    (1024, 256, 150, 3, 10, 3),    # Q = 4, the headline frame size, music-mode schedule
    (1024, 256, 40, 5, 3, 1),      # longer look-ahead
    (1024, 256, 60, 5, 2, 1),      # ... with few iterations: the 16-frame ring is exactly full
    (1024, 256, 33, 0, 4, 1),      # no look-ahead
    (1024, 512, 90, 3, 6, 2),      # Q = 2
    (512, 64, 60, 2, 5, 2),        # Q = 8
    (256, 64, 5, 3, 4, 1),         # fewer frames than the window
    (256, 64, 1, 3, 2, 1),         # a single frame
    (512, 128, 40, 9, 2, 1),       # long look-ahead: few sweep slots among the 64 lanes of the wave-per-tap-group layout
])
@pytest.mark.parametrize("layout", ["2", "3", "4"])
def test_serial_variant_is_bit_identical_to_generic(fsize, fshift, T, LA, iters, B, layout, monkeypatch):
    """The lane layouts of the LDS engine (2: 2Q lanes per bin; 3: one wave per tap group with the projection wave a step behind;
    4: the same roles on two-step windows of aligned cells, even lag, run-time ring -- the default; LWS_ONLINE_LAYOUT forces one
    where it fits, else another runs)."""
    monkeypatch.setenv("LWS_ONLINE_LAYOUT", layout)
    rng = np.random.default_rng(fsize + T)
    p = lws_amd.lws(fsize, fshift, mode="music")
    F = fsize // 2 + 1
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    if B > 1:
        S[1] = np.abs(S[1])                       # magnitudes with zero phase, as run_lws feeds them
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    ref, name = _online(F, W, S, thr, LA, fsize / fshift, force_generic=True)
    assert name == "generic_fp32"
    monkeypatch.setenv("LWS_ONLINE_SERIAL_TAPS", "1")
    out, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    assert np.array_equal(out, ref)
    # the production variant only re-associates the sum over frames: same magnitudes
    monkeypatch.delenv("LWS_ONLINE_SERIAL_TAPS")
    prod, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    assert np.abs(np.abs(prod) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()


@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8"])
@pytest.mark.parametrize("T", [1, 2, 3, 7, 24])
@pytest.mark.parametrize("LA", [0, 1, 3, 5])
@pytest.mark.parametrize("layout", ["2", "3", "4"])
def test_small_shapes_vs_oracle(tag, T, LA, layout, oracle, monkeypatch):
    monkeypatch.setenv("LWS_ONLINE_LAYOUT", layout)
    h, g = load_golden("helpers.npz"), load_golden("wrappers.npz")
    W = (h[f"W_{tag}"], h[f"W_ai_{tag}"], h[f"W_af_{tag}"])
    fsize, fshift = [int(v) for v in tag.split("_")]
    S = g[f"S_{tag}"][:T]
    F = S.shape[1]
    thr = [0.6, 0.2, 0.0]
    ref = oracle.online_lws(S, *W, thr, LA, fshift)
    out, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    err = np.abs(out - ref)
    scale = np.mean(np.abs(S))
    assert np.median(err) < 2e-6 * scale and np.linalg.norm(err) < 5e-3 * np.linalg.norm(ref)
    assert np.abs(np.abs(out) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()


@pytest.mark.parametrize("fsize,fshift,T,LA,iters", [(1024, 256, 10, 3, 2), (1024, 256, 16, 3, 3), (512, 128, 12, 3, 3),
                                                    (1024, 512, 12, 3, 3), (512, 64, 10, 2, 2)])
@pytest.mark.parametrize("layout", ["2", "3", "4"])
def test_production_variant_vs_oracle_at_full_frame_sizes(fsize, fshift, T, LA, iters, layout, oracle, monkeypatch):
    """The production tap order (per-lane / per-wave partial sums, windows in registers, the projection wave's late terms)
    against the fp64 oracle at the frame sizes of the BASELINE configs, on runs short enough (a dozen frames, 2-3
    iterations) for fp32 rounding to stay rounding: values, not just magnitudes."""
    monkeypatch.setenv("LWS_ONLINE_LAYOUT", layout)
    rng = np.random.default_rng(fsize + T)
    p = lws_amd.lws(fsize, fshift, mode="music")
    F = fsize // 2 + 1
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    ref = oracle.online_lws(S, p.W, p.W_ai, p.W_af, thr, LA, fshift)
    out, name = _online(F, (p.W, p.W_ai, p.W_af), S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    err, scale = np.abs(out - ref), np.mean(np.abs(S))
    assert np.median(err) < 1e-6 * scale and np.linalg.norm(err) < 2e-4 * np.linalg.norm(ref)
    assert np.abs(np.abs(out) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()


@pytest.mark.parametrize("fsize,fshift,L,T,LA,iters,B", [(1024, 256, 3, 60, 3, 6, 2), (512, 128, 1, 40, 3, 4, 1),
                                                         (1024, 512, 4, 50, 2, 5, 2), (512, 64, 3, 40, 2, 4, 1),
                                                         (1000, 250, 2, 45, 3, 5, 1), (2048, 512, 3, 12, 3, 3, 1)])
def test_narrower_stencils(fsize, fshift, L, T, LA, iters, B, oracle, monkeypatch):
    """`lws(..., L=3, mode='music')` (lws.pyx:379 takes any L): stencils narrower than the kernel's run as L = 5 with zero
    weights for the taps they do not have, on the caller's narrower pad columns.  Serial-taps variant bit-identical to the
    generic engine; production variant: same magnitudes, and the oracle's values on a short run."""
    rng = np.random.default_rng(fsize + T + L)
    p = lws_amd.lws(fsize, fshift, L=L, mode="music")
    F = fsize // 2 + 1
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    ref, name = _online(F, W, S, thr, LA, fsize / fshift, force_generic=True)
    assert name == "generic_fp32"
    monkeypatch.setenv("LWS_ONLINE_SERIAL_TAPS", "1")
    out, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    assert np.array_equal(out, ref)
    monkeypatch.delenv("LWS_ONLINE_SERIAL_TAPS")
    prod, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    assert np.abs(np.abs(prod) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()
    Ts = min(T, 10)
    short, _ = _online(F, W, S[0, :Ts], thr[:2], LA, fsize / fshift)
    o = oracle.online_lws(S[0, :Ts], *W, thr[:2], LA, fshift)
    err = np.abs(short - o)
    assert np.median(err) < 2e-6 * np.mean(np.abs(S)) and np.linalg.norm(err) < 5e-3 * np.linalg.norm(o)


@pytest.mark.parametrize("fsize,fshift,T,iters,LA", [(4096, 1024, 12, 2, 3), (3000, 750, 14, 3, 3), (4096, 2048, 16, 3, 2), (4000, 1000, 11, 2, 3)])
def test_frames_of_4096_points(fsize, fshift, T, iters, LA, oracle, monkeypatch):
    """Frames too long for the LDS ring to hold their target magnitudes and the step table beside the values (a ring of eight
    2060-column frames is 132 KB): the kernel's BIG variant reads the magnitudes from the caller's buffer and computes the table
    entries, with a sweep lag long enough for eight ring frames.  Against the fp64 oracle on short runs (values), same magnitudes as
    the generic engine; the serial-taps verification variant has no such build and runs on the generic engine."""
    rng = np.random.default_rng(fsize + T)
    p = lws_amd.lws(fsize, fshift, mode="music")
    F = fsize // 2 + 1
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    out, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32"
    ref = oracle.online_lws(S, *W, thr, LA, fshift)
    gen, name = _online(F, W, S, thr, LA, fsize / fshift, force_generic=True)
    assert name == "generic_fp32"
    err, scale = np.abs(out - ref), np.mean(np.abs(S))
    gen_l2, gen_med = np.linalg.norm(gen - ref) / np.linalg.norm(ref), np.median(np.abs(gen - ref))
    assert np.median(err) < max(1e-6 * scale, 3 * gen_med), (np.median(err), gen_med)
    assert np.linalg.norm(err) < max(1e-3, 3 * gen_l2) * np.linalg.norm(ref), (np.linalg.norm(err) / np.linalg.norm(ref), gen_l2)
    assert np.abs(np.abs(out) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()
    monkeypatch.setenv("LWS_ONLINE_SERIAL_TAPS", "1")
    ser, name = _online(F, W, S, thr, LA, fsize / fshift)
    # (where the frames still fit the ring with their magnitudes -- 1501 bins, or Q = 2 -- the verification variant runs on the LDS
    #  kernel; either way: the generic engine's bits)
    assert name in ("generic_fp32", "online_lds_fp32") and np.array_equal(ser, gen)
    if fsize == 4096 and fshift == 1024:
        assert name == "generic_fp32"


def test_more_than_eight_frames_per_row_go_to_the_team_engine(monkeypatch):
    """More than 8 frames per stencil row has no LDS engine: the team engine (lws_team.hip, tests/test_gpu_team.py) in fp32 and in fp64,
    the generic engine with LWS_NO_TEAM=1.  (Q = 3, 5, 6, 7 and fractional Q: the table-twiddle variant of the fourth layout,
    tests/test_gpu_tw.py; fp64 plans of Q in {2,3,4,8}: lws_online64.hip, tests/test_gpu_online64.py.)"""
    rng = np.random.default_rng(0)
    p = lws_amd.lws(144, 16, mode="music")           # Q = 9
    S = rng.standard_normal((9, 73)) + 1j * rng.standard_normal((9, 73))
    out, name = _online(73, (p.W, p.W_ai, p.W_af), S, [0.5, 0.1], 3, 9.0)
    assert name == "team_online_fp32"
    out64, name = _online(73, (p.W, p.W_ai, p.W_af), S, [0.5, 0.1], 3, 9.0, precision="fp64")
    assert name == "team_online_ordered_fp64"          # (fp64 plans: the order-exact kernel)
    monkeypatch.setenv("LWS_NO_TEAM", "1")
    gen, name = _online(73, (p.W, p.W_ai, p.W_af), S, [0.5, 0.1], 3, 9.0)
    assert name == "generic_fp32"
    gen64, name = _online(73, (p.W, p.W_ai, p.W_af), S, [0.5, 0.1], 3, 9.0, precision="fp64")
    assert name == "generic_fp64"
    monkeypatch.delenv("LWS_NO_TEAM")
    assert np.array_equal(out64, gen64)
    assert np.linalg.norm(out - gen) < 1e-3 * np.linalg.norm(gen)
    p = lws_amd.lws(64, 16, mode="music")
    S = rng.standard_normal((9, 33)) + 1j * rng.standard_normal((9, 33))
    out, name = _online(33, (p.W, p.W_ai, p.W_af), S, [0.5, 0.1], 3, 4.0, precision="fp64")
    assert name == "online_lds_fp64"


@pytest.mark.parametrize("iters,T", [(1, 30), (3, 14), (10, 8)])
def test_wide_frames_stay_on_the_lds_engine(iters, T, oracle, monkeypatch):
    """2048-point frames (BASELINE config 5's frame size, lws(2048,512, mode='music')): the run-time ring of the fourth layout
    holds them -- with one iteration per frame by lengthening the lag between sweeps -- where the 16-frame rings of the other
    layouts sent them to the generic engine.  Serial-taps variant bit-identical to the generic engine, production variant
    against the fp64 oracle (short runs: values)."""
    rng = np.random.default_rng(iters)
    p = lws_amd.lws(2048, 512, mode="music")
    S = rng.standard_normal((T, 1025)) + 1j * rng.standard_normal((T, 1025))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    out, name = _online(1025, W, S, thr, 3, 4.0)
    assert name == "online_lds_fp32"
    ref = oracle.online_lws(S, *W, thr, 3, 512)
    gen, name = _online(1025, W, S, thr, 3, 4.0, force_generic=True)
    assert name == "generic_fp32"
    err, scale = np.abs(out - ref), np.mean(np.abs(S))
    # (30 frames of one iteration each amplify fp32 rounding: the order-exact generic fp32 engine is the yardstick there)
    gen_l2, gen_med = np.linalg.norm(gen - ref) / np.linalg.norm(ref), np.median(np.abs(gen - ref))
    assert np.median(err) < max(1e-6 * scale, 3 * gen_med), (np.median(err), gen_med)
    assert np.linalg.norm(err) < max(1e-3, 3 * gen_l2) * np.linalg.norm(ref), gen_l2
    assert np.abs(np.abs(out) - np.abs(ref)).max() < 2e-6 * np.abs(S).max()
    monkeypatch.setenv("LWS_ONLINE_SERIAL_TAPS", "1")
    ser, name = _online(1025, W, S, thr, 3, 4.0)
    assert name == "online_lds_fp32" and np.array_equal(ser, gen)


def test_config3_online_stage_at_full_size(monkeypatch):
    """BASELINE config 3's online stage at its full per-spectrogram size -- 500 frames x 513 bins, 10 iterations, look-ahead 3,
    magnitudes with zero phase as run_lws feeds them -- on the default (fourth) layout: the verification variant reproduces the
    order-exact generic engine bit for bit over all 5500 sweeps (schedule, frame ring, slots, step table), and the production
    variant -- which only re-associates sums -- keeps every magnitude and reaches the same consistency."""
    rng = np.random.default_rng(33)
    p = lws_amd.lws(1024, 256, mode="music")
    B, T, F = 2, 500, 513
    S = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
    thr = lws_amd.get_thresholds(10, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    gen, name = _online(F, W, S, thr, 3, 4.0, force_generic=True)
    assert name == "generic_fp32"
    monkeypatch.setenv("LWS_ONLINE_SERIAL_TAPS", "1")
    ser, name = _online(F, W, S, thr, 3, 4.0)
    assert name == "online_lds_fp32" and np.array_equal(ser, gen)
    monkeypatch.delenv("LWS_ONLINE_SERIAL_TAPS")
    prod, name = _online(F, W, S, thr, 3, 4.0)
    assert name == "online_lds_fp32"
    assert np.abs(np.abs(prod) - np.abs(S)).max() < 2e-6 * np.abs(S).max()
    for b in range(B):
        assert abs(p.get_consistency(prod[b]) - p.get_consistency(gen[b])) < 0.03
