"""GPU (-m gpu): the table-twiddle builds of the systolic kernel (lws_systolic.hip -DLWS_TW=1: namespaces lws::tw, lws::tw_half)
-- Q = 3 (the reference's LWSanyQ) and the general weights create_weights builds when the hop does not divide the frame
(lws.pyx:164-168: Q' = N rows, LWSfractionalQ, lwslib.cpp:376-467; e.g. 25 ms frames every 10 ms = lws(400, 160)) -- against the
oracle, whose fractional path is pinned by tests/golden/general_weights.npz (with the reference's out-of-bounds weight row N
defined as row 0).  Same bars as tests/test_gpu_systolic.py::run_case; the fp64 generic engine pins the
schedule for every case."""
import os

import numpy as np
import pytest

import lws_amd
from conftest import load_golden
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def tw_case(oracle, fsize, fshift, T, thr, seed, B=2, scale=(1.0, 40.0), L=5, use_simplifications=True, expect="tw"):
    p = lws_amd.lws(fsize, fshift, L=L, use_simplifications=use_simplifications)
    F = fsize // 2 + 1
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    S *= np.asarray(scale)[:B, None, None]
    plan = _capi.Plan(F, p.W)
    out = plan.batch(S, thr)
    name = plan.last_kernel()["name"]
    assert name.startswith("systolic") and name.endswith("_" + expect), name
    r64 = -(-fsize // fshift) >= 5          # five or more frames per stencil row: the 64-step ring (frames of up to 513 bins)
    period = fsize // np.gcd(fsize, fshift)  # of the twiddles, in bins: the table of the two-slots-per-wave build holds 22 rows + 8
    half = F <= 257 and period <= 22
    if expect == "tw" and not r64:
        assert ("_half_" in name) == half and ("_wide_" in name) == (F > 513), name
    if expect == "tw" and half and not r64:
        # the build with two sweep slots per wave does the arithmetic of the one-slot build in the same order: identical bits
        os.environ["LWS_SYSTOLIC_NO_SHORT"] = "1"
        try:
            narrow = _capi.Plan(F, p.W)
        finally:
            del os.environ["LWS_SYSTOLIC_NO_SHORT"]
        assert np.array_equal(narrow.batch(S, thr), out) and "_half_" not in narrow.last_kernel()["name"], narrow.last_kernel()
        narrow.close()
    p64 = _capi.Plan(F, p.W, precision="fp64")
    worst = 0.0
    for b in range(B):
        ref = oracle.batch_lws(S[b], p.W, thr)
        assert np.abs(p64.batch(S[b], thr) - ref).max() < 1e-8
        mean = np.mean(np.abs(S[b]))
        d = np.abs(out[b] - ref)
        worst = max(worst, rel_l2(out[b], ref))
        assert rel_l2(out[b], ref) < 3e-3, (fsize, fshift, T, b, rel_l2(out[b], ref))
        assert np.median(d) < 2e-6 * mean
        assert np.abs(np.abs(out[b]) - np.abs(S[b])).max() < 2e-6 * np.abs(S[b]).max()
    p64.close(); plan.close()
    return out, name


THR = [0.5, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]


@pytest.mark.parametrize("fsize,fshift,T", [(48, 16, 70), (96, 32, 131), (384, 128, 37), (768, 256, 66), (1008, 336, 40), (60, 20, 64),
                                            (1020, 340, 33), (996, 332, 37), (984, 328, 40)])
def test_q3_summarised_weights(oracle, fsize, fshift, T):
    """hop = a third of the frame: Q = 3, summarised weights W[3][3][L+1] with twiddle exp(2 pi j (bin mod 3) r / 3) -- the
    reference's LWSanyQ (lws.pyx:252-253, lwslib.cpp:283-373).  Frame ends at every phase of a block."""
    tw_case(oracle, fsize, fshift, T, THR, seed=fsize + T)


@pytest.mark.parametrize("fsize,fshift,T", [(400, 160, 70), (400, 160, 131), (512, 160, 66), (1000, 400, 37), (1024, 384, 40), (600, 250, 65),
                                            (80, 32, 70), (1024, 320, 21), (644, 230, 50), (1012, 368, 33),
                                            # a hop above half the frame: two frames per stencil row, general weights
                                            (512, 300, 66), (1024, 640, 37), (400, 240, 70), (64, 40, 131), (2048, 1280, 40)])
def test_fractional_q_general_weights(oracle, fsize, fshift, T):
    """A hop that does not divide the frame: create_weights returns one weight row per bin (Q' = N) and batch_lws dispatches to
    LWSfractionalQ (lws.pyx:246-247).  25 ms / 10 ms speech framing is lws(400, 160): Q = 3 frames, twiddle period 5 bins."""
    out, name = tw_case(oracle, fsize, fshift, T, THR, seed=fsize + T)
    p = lws_amd.lws(fsize, fshift)
    assert p.W.shape[0] == fsize and not float(p.Q).is_integer()


@pytest.mark.parametrize("fsize,fshift,T", [(2048, 768, 40), (1536, 512, 70), (2000, 800, 33), (2044, 700, 131), (1200, 480, 66), (2048, 640, 21),
                                            (1980, 660, 37)])
def test_wide_frames(oracle, fsize, fshift, T):
    """Frames of 515 .. 1025 bins (a 2048-point window every 768 samples: Q = 3, Qfloat = 2.67): the table-twiddle code under the
    build with two waves per sweep slot (lws::tw_wide)."""
    tw_case(oracle, fsize, fshift, T, THR, seed=fsize + T)


@pytest.mark.parametrize("fsize,fshift", [(64, 16), (1024, 256), (1024, 512), (1024, 128), (2048, 512), (512, 128)])
def test_general_weights_of_an_integer_q_use_the_static_builds(oracle, fsize, fshift):
    """use_simplifications=False with a hop that divides the frame: N weight rows that are periodic copies of the Q summarised ones
    (SURVEY.md probe 3-5).  The library verifies that on every row and serves the plan with the ordinary builds."""
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, use_simplifications=False)
    assert p.W.shape[0] == fsize
    rng = np.random.default_rng(fsize)
    S = rng.standard_normal((2, 37, F)) + 1j * rng.standard_normal((2, 37, F))
    thr = [0.5, 0.0, 0.0]
    plan = _capi.Plan(F, p.W)
    out = plan.batch(S, thr)
    name = plan.last_kernel()["name"]
    assert name.startswith("systolic") and not name.endswith("_tw"), name
    ps = lws_amd.lws(fsize, fshift)
    assert np.array_equal(out, _capi.Plan(F, ps.W).batch(S, thr))      # the summarised tensor gives the same bits
    for b in range(2):
        assert rel_l2(out[b], oracle.batch_lws(S[b], p.W, thr)) < 3e-3


def test_reference_goldens_of_the_fractional_path():
    """tests/golden/general_weights.npz: LWSfractionalQ of the reference itself on lws(32, 12) weights (Q = 3, Qfloat = 2.67)
    and lws(32, 8) with use_simplifications=False, through the table-twiddle / static builds in fp32."""
    g = load_golden("general_weights.npz")
    for tag, want_tw in (("32_12", True), ("32_8", False)):
        fsize, fshift, T, F, Q, L, LA = [int(v) for v in g[f"meta_{tag}"]]
        S, thr = g[f"S_{tag}"], float(g[f"thr_{tag}"][0])
        mean = np.mean(np.abs(S))
        plan = _capi.Plan(F, g[f"W_{tag}"])
        out = plan.batch(S, [thr / mean, 0.0])
        name = plan.last_kernel()["name"]
        assert name.startswith("systolic") and name.endswith("_tw") == want_tw, name
        ref = g[f"batch_{tag}"][Q - 1:Q - 1 + T, L:L + F]
        assert rel_l2(out, ref) < 3e-3 and np.median(np.abs(out - ref)) < 2e-6 * mean, (tag, rel_l2(out, ref))
        plan.close()


@pytest.mark.parametrize("n_it", [1, 6, 7, 8, 15, 22])
def test_sweep_counts_around_slot_groups(oracle, n_it):
    tw_case(oracle, 400, 160, 21, np.linspace(0.8, 0.0, n_it), seed=100 + n_it)
    tw_case(oracle, 1000, 400, 21, np.linspace(0.8, 0.0, n_it), seed=200 + n_it)


@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_narrower_stencils(oracle, L):
    """L < 5 runs on the L = 5 build with zero weights for the taps the caller's tensors do not have."""
    tw_case(oracle, 400, 160, 40, THR, seed=L, L=L)
    tw_case(oracle, 768, 256, 40, THR, seed=10 + L, L=L)


def test_what_the_table_builds_do_not_take():
    """More than 8 frames per stencil row, Q >= 5 above 513 bins, L > 5 with table twiddles: the band engine (round 6; generic before)."""
    import warnings
    for fsize, fshift, L in ((1008, 112, 5), (2000, 400, 5), (400, 160, 7), (1000, 200, 6)):
        p = lws_amd.lws(fsize, fshift, L=L)
        with warnings.catch_warnings():
            warnings.simplefilter("error")                      # (a plan on the generic engine warns)
            p.batch_lws(np.ones((4, fsize // 2 + 1)), thresholds=[0.0])
        assert p.plan().last_kernel()["name"] == "band_fp32", (fsize, fshift)


@pytest.mark.parametrize("fsize,fshift,B,T,iters", [(400, 160, 3, 300, 30), (1000, 400, 2, 200, 30), (768, 256, 5, 260, 16), (2048, 768, 3, 150, 20),
                                                    (1000, 200, 2, 300, 9), (1024, 160, 3, 200, 7)])
def test_workgroups_sharing_a_spectrogram_change_nothing(fsize, fshift, B, T, iters, monkeypatch):
    rng = np.random.default_rng(B * T)
    F = fsize // 2 + 1
    S = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex128)
    S[0] *= 30.0
    thr = lws_amd.get_thresholds(iters, 3.0, 0.15, 1)
    p = lws_amd.lws(fsize, fshift)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "1")
    ref = p.plan().batch(S, thr)
    assert p.plan().last_kernel()["name"].endswith("_tw")
    for nwg in ("2", "3", "4"):
        monkeypatch.setenv("LWS_SYSTOLIC_NWG", nwg)
        assert np.array_equal(p.plan().batch(S, thr), ref), nwg
    monkeypatch.delenv("LWS_SYSTOLIC_NWG")
    assert np.array_equal(p.plan().batch(S, thr), ref)


@pytest.mark.parametrize("fsize,fshift,T", [(400, 160, 150), (1000, 400, 150), (768, 256, 100), (2048, 768, 100), (1000, 200, 150), (1024, 160, 100)])
def test_stalled_waves_change_nothing(fsize, fshift, T, monkeypatch):
    rng = np.random.default_rng(T)
    F = fsize // 2 + 1
    S = np.abs(rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))).astype(np.complex128)
    thr = np.zeros(9)
    p = lws_amd.lws(fsize, fshift)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "1")
    ref = p.plan().batch(S, thr)
    for role in range(8):
        for pair in (1, 3, 5, 7):
            monkeypatch.setenv("LWS_SYSTOLIC_STRESS", str((1 << role) | (pair << 16)))
            assert np.array_equal(p.plan().batch(S, thr), ref), (role, pair)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "3")
    for mask, pair in ((0x40, 1), (0x80, 1), (0x0f, 3), (0x30, 7)):
        monkeypatch.setenv("LWS_SYSTOLIC_STRESS", str(mask | (pair << 16)))
        assert np.array_equal(p.plan().batch(S, thr), ref), (mask, pair)


def test_speech_framing_end_to_end():
    """lws(400, 160) -- 25 ms frames every 10 ms at 16 kHz -- through the class: stft, run_lws on magnitudes, consistency up."""
    p = lws_amd.lws(400, 160, batch_iterations=60, batch_alpha=20)
    x = np.random.default_rng(0).standard_normal(16000)
    X = p.stft(x)
    Y = p.run_lws(np.abs(X))
    assert p.plan().last_kernel()["name"] == "systolic_half_q3_l5_tw"
    assert Y.dtype == np.complex128 and np.abs(np.abs(Y) - np.abs(X)).max() < 1e-6 * np.abs(X).max()
    assert p.get_consistency(Y) > p.get_consistency(np.abs(X).astype(complex)) + 5.0
    # fp16 storage and device-resident calls take the same build
    import torch
    ph = lws_amd.lws(400, 160, storage="fp16")
    t = torch.from_numpy(np.abs(X)[None].astype(np.complex64)).cuda()
    ph.plan().batch_dev(t.data_ptr(), 1, X.shape[0], np.zeros(20), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert ph.plan().last_kernel()["name"] == "systolic_half_q3_l5_tw_f16"
    assert abs(ph.get_consistency(t.cpu().numpy()[0].astype(np.complex128)) - p.get_consistency(p.batch_lws(np.abs(X), thresholds=np.zeros(20)))) < 0.3


@pytest.mark.parametrize("fsize,fshift,T,n_it", [(400, 160, 40, 2), (1000, 400, 25, 1), (512, 160, 33, 3), (2048, 768, 12, 2), (32, 12, 14, 2)])
def test_nofuture_sweeps_with_general_weights_run_on_the_lds_engine(oracle, fsize, fshift, T, n_it):
    """NoFuture_LWSfractionalQ (lwslib.cpp:693-764): weight row = bin.  The rows of create_weights' general tensors repeat with
    period frame / gcd(frame, hop); the LDS engine keeps one period of them and indexes row (bin mod period).  Against the oracle
    (fp32 bars) and against the order-exact generic engine."""
    p = lws_amd.lws(fsize, fshift)
    F = fsize // 2 + 1
    rng = np.random.default_rng(fsize + T)
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    S[1] *= 30.0
    thr = np.linspace(0.5, 0.0, n_it)
    plan = _capi.Plan(F, p.W, p.W_ai, p.W_af)
    out = plan.nofuture(S, thr, wsel=_capi.LWS_W_AI)
    assert plan.last_kernel()["name"] == "nofuture_lds_fp32", plan.last_kernel()
    gen = _capi.Plan(F, p.W, p.W_ai, p.W_af, force_generic=True)
    outg = gen.nofuture(S, thr, wsel=_capi.LWS_W_AI)
    assert gen.last_kernel()["name"] == "generic_fp32"
    p64 = _capi.Plan(F, p.W, p.W_ai, p.W_af, precision="fp64")
    for b in range(2):
        ref = oracle.nofuture_lws(S[b], p.W_ai, thr, compat=False)
        assert rel_l2(p64.nofuture(S[b], thr, wsel=_capi.LWS_W_AI), ref) < 1e-8       # the schedule, in fp64 (1e-10: the recursion amplifies fp64 rounding too)
        # A no-future sweep is a recursion along the frames (frame m from frames m-1 .. m-Q+1) that amplifies rounding about
        # threefold every five frames at these sizes -- the oracle itself moves by 1e-2 over 40 frames when its weights are rounded
        # to fp32 (tools/nf_diag.py) -- so: value-level on the first frames, and overall no worse than the order-exact engine
        e_lds, e_gen = rel_l2(out[b], ref), rel_l2(outg[b], ref)
        assert rel_l2(out[b][:8], ref[:8]) < 1e-4 and rel_l2(outg[b][:8], ref[:8]) < 1e-4, (fsize, b)
        assert e_lds < 4 * e_gen + 1e-3 and e_lds < 0.2, (fsize, b, e_lds, e_gen)
        assert np.abs(np.abs(out[b]) - np.abs(S[b])).max() < 2e-6 * np.abs(S[b]).max()
    plan.close(); gen.close(); p64.close()


def _online(F, W, S, thr, LA, qdiv, **kw):
    plan = _capi.Plan(F, *W, **kw)
    out = plan.online(S, thr, LA, qdiv)
    name = plan.last_kernel()["name"]
    plan.close()
    return out, name


@pytest.mark.parametrize("fsize,fshift,T,LA,iters", [(48, 16, 24, 3, 3), (48, 16, 7, 0, 2), (96, 32, 16, 1, 3), (768, 256, 12, 3, 3), (400, 160, 14, 3, 3),
                                                    (400, 160, 30, 5, 2), (512, 160, 12, 3, 3), (1024, 384, 10, 3, 2), (1000, 400, 12, 2, 3),
                                                    (80, 32, 20, 3, 3), (60, 20, 9, 3, 2),
                                                    (512, 300, 12, 3, 3), (64, 40, 20, 2, 2),
                                                    # five to eight frames per stencil row: ten to sixteen waves
                                                    (80, 16, 16, 3, 3), (1000, 200, 10, 3, 2), (768, 128, 12, 2, 3), (896, 128, 9, 3, 2), (1024, 160, 10, 3, 2),
                                                    (1024, 176, 12, 1, 3), (96, 16, 20, 0, 2),
                                                    # eight frames per row with table twiddles (more than seven hops per frame, not eight)
                                                    (960, 128, 10, 3, 2), (1024, 144, 10, 3, 2), (96, 13, 20, 3, 2)])
def test_online_sweeps_on_the_lds_engine(fsize, fshift, T, LA, iters, oracle):
    """TF_RTISI_LA with Q = 3 and with the general weights of a fractional Q (Asym_UpdatePhaseanyQ / Asym_UpdatePhasefractionalQ,
    lwslib.cpp:1129-1421): the fourth layout of the online LDS engine with its twiddles from a table (k_online4<..., TWT>).  Short
    runs -- values against the fp64 oracle, as for the static builds (tests/test_gpu_online.py) -- and the same magnitudes as the
    order-exact generic engine on every bin."""
    rng = np.random.default_rng(fsize + T)
    p = lws_amd.lws(fsize, fshift, mode="music")
    F = fsize // 2 + 1
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    ref = oracle.online_lws(S, *W, thr, LA, fshift)
    out, name = _online(F, W, S, thr, LA, fsize / fshift)
    assert name == "online_lds_fp32", name
    gen, gname = _online(F, W, S, thr, LA, fsize / fshift, force_generic=True)
    assert gname == "generic_fp32"
    err, gerr, scale = np.abs(out - ref), np.abs(gen - ref), np.mean(np.abs(S))
    # (the longer runs already show the stage's own amplification of rounding -- the order-exact fp32 engine is the yardstick there)
    assert rel_l2(out[:8], ref[:8]) < 1e-4 and np.median(err[:8]) < 2e-6 * scale, (rel_l2(out[:8], ref[:8]), np.median(err[:8]) / scale)
    assert np.median(err) < max(2e-6 * scale, 10 * np.median(gerr)), (np.median(err) / scale, np.median(gerr) / scale)
    assert rel_l2(out, ref) < max(1e-3, 5 * rel_l2(gen, ref)), (rel_l2(out, ref), rel_l2(gen, ref))
    assert np.abs(np.abs(out) - np.abs(gen)).max() < 2e-6 * np.abs(S).max()
    # the first frames against the order-exact fp32 engine: the same arithmetic in another order (a dropped or mis-twiddled tap --
    # frame rho-1's last column is ~1 % of a bin's sum -- is two orders of magnitude above this)
    assert rel_l2(out[:3], gen[:3]) < 3e-5, rel_l2(out[:3], gen[:3])


def test_music_mode_with_speech_framing(oracle):
    """lws(400, 160, mode='music').run_lws: no-future -> online -> batch, all three on their fast engines.  From zero-phase magnitudes
    rounding decides many phases in the very first sweep (tests/test_gpu_parity.py), so the pipeline is compared with the oracle's
    by the consistency it reaches; the stages' values are pinned above on well-conditioned input."""
    p = lws_amd.lws(400, 160, mode="music", online_iterations=4, batch_iterations=30, batch_alpha=5.0, nofuture_q4_compat=False)
    rng = np.random.default_rng(4)
    M = np.abs(rng.standard_normal((80, 201)) + 1j * rng.standard_normal((80, 201)))
    names = []
    s0 = p.nofuture_lws(M); names.append(p.plan().last_kernel()["name"])
    s1 = p.online_lws(s0); names.append(p.plan().last_kernel()["name"])
    s2 = p.batch_lws(s1); names.append(p.plan().last_kernel()["name"])
    assert names == ["nofuture_lds_fp32", "online_lds_fp32", "systolic_half_q3_l5_tw"], names
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # nothing lands on the generic engine any more
        out = p.run_lws(M)
    r0 = oracle.nofuture_lws(M, p.W_ai, lws_amd.get_thresholds(1, 1, 0.1, 1), compat=False)
    r1 = oracle.online_lws(r0, p.W, p.W_ai, p.W_af, lws_amd.get_thresholds(4, 1, 0.1, 1), 3, 160)
    r2 = oracle.batch_lws(r1, p.W, lws_amd.get_thresholds(30, 5.0, 0.1, 1))
    assert np.abs(np.abs(out) - M).max() < 2e-6 * M.max() and np.abs(np.abs(s2) - M).max() < 2e-6 * M.max()
    # (80 x 201 bins from a zero-phase start: the trajectories of fp32 and fp64 part in the first frames and the consistency of so
    # small a spectrogram moves by a dB or two with them; what must hold is that every stage reaches the oracle's level)
    cons = [(p.get_consistency(mine), p.get_consistency(ref)) for mine, ref in ((s0, r0), (s1, r1), (out, r2))]
    assert all(a > b - 1.0 and a < b + 3.0 for a, b in cons), cons
    assert p.get_consistency(out) > p.get_consistency(M.astype(complex)) + 4.0


@pytest.mark.parametrize("fsize,fshift,T", [(80, 16, 70), (1000, 200, 37), (960, 192, 66), (96, 16, 131), (768, 128, 40), (1020, 170, 33),
                                            (112, 16, 70), (896, 128, 37), (1008, 144, 65), (1024, 160, 40), (1024, 176, 21), (1000, 150, 37),
                                            (1024, 192, 33), (100, 20, 64), (1012, 184, 30), (1024, 128, 37)])
def test_five_to_eight_frames_per_stencil_row(oracle, fsize, fshift, T):
    """ceil(frame / hop) in 5..8 -- Q = 5, 6, 7 (the reference's LWSanyQ) and fractional Q above 4 (LWSfractionalQ; lws(1024, 160):
    Qfloat = 6.4) -- on the 64-step ring of the Q = 8 build with table twiddles (lws::tw_q8: the Q = 8 kernel, the frame pairs the plan
    does not have masked out; Q = 5 and 6 on the same kernel with a 40- / 48-step ring and three sweep slots, lws::tw_q5 / tw_q6).
    lws(1024, 128) itself stays on the static Q = 8 build."""
    q = -(-fsize // fshift)
    static = fsize % fshift == 0 and q == 8
    out, name = tw_case(oracle, fsize, fshift, T, THR, seed=fsize + T, expect="hann" if static else "tw")
    ring = {5: 40, 6: 48}.get(q, 64)            # (round 5: exactly 5 / 6 frames per row run on a ring of their own depth, lws::tw_q5 / tw_q6)
    assert ("_r%d_q%d_" % (ring, q) in name) == (not static), name


@pytest.mark.parametrize("fsize,fshift,T", [(1000, 200, 37), (112, 16, 70), (100, 20, 64), (400, 160, 70), (768, 256, 40)])
def test_results_do_not_depend_on_stale_device_memory(oracle, fsize, fshift, T):
    """The scratch of a plan comes from hipMalloc as it is.  Entries of the kernel's layout that no frame owns (frames beyond the padded
    spectrogram, as the Nyquist lanes of lws::tw_q8 see them for the frame offsets a plan with Q < 8 does not have) must never reach a
    result, not even multiplied by a zero weight: fill the device's free memory with NaNs, give it back, and run."""
    import torch
    for fill in (float("nan"), float("inf"), 1e30):
        x = torch.full((1 << 28,), fill, dtype=torch.float32, device="cuda")      # 1 GiB
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
        tw_case(oracle, fsize, fshift, T, THR, seed=fsize + T, expect="tw")


@pytest.mark.parametrize("fsize,fshift", [(512, 128), (256, 128), (1024, 128)])
def test_table_twiddles_where_static_ones_would_do(fsize, fshift, monkeypatch):
    """LWS_ONLINE_TABLE_TWIDDLES=1 (read when the plan is made) runs k_online4<..., TWT> on hop = frame / 2, 4, 8 too: the same sums with
    the twiddles from the table (cos / sin rounded to fp32: 6e-17 where the static variant has an exact zero, which rarely moves a bit)
    -- the first frames agree to rounding, the magnitudes everywhere.  This is how the table's cost per step is measured (+7 %)."""
    rng = np.random.default_rng(fsize)
    F, T, LA = fsize // 2 + 1, 14, 3
    p = lws_amd.lws(fsize, fshift, mode="music")
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(3, 1.0, 0.1, 1)
    W = (p.W, p.W_ai, p.W_af)
    a, name_a = _online(F, W, S, thr, LA, fsize / fshift)
    monkeypatch.setenv("LWS_ONLINE_TABLE_TWIDDLES", "1")
    b, name_b = _online(F, W, S, thr, LA, fsize / fshift)
    assert name_a == name_b == "online_lds_fp32"
    assert rel_l2(b[:3], a[:3]) < 3e-5, rel_l2(b[:3], a[:3])
    assert np.abs(np.abs(a) - np.abs(b)).max() < 2e-6 * np.abs(S).max()
