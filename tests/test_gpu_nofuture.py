"""GPU (-m gpu): the LDS-resident no-future kernel (lws_nofuture.hip) against the order-exact generic engine -- which the
goldens pin to the reference (tests/test_gpu_parity.py, tests/test_gpu_compat.py) -- over both addressing modes
(NoFuture_LWSanyQ semantics and the shipped NoFuture_LWSQ4's flat offset, lwslib.cpp:538-617), several Q and L, multi-sweep
schedules and the tiny frames where the compat rounds degenerate; plus the oracle in fp64.  The kernel's verification variant
(LWS_NOFUTURE_SERIAL_TAPS=1: one lane sums the taps of a bin in the generic engine's order) must match bit for bit in fp32;
the production variant (eight lanes per bin, partial sums combined across lanes) re-associates the sum."""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def weights(fsize, fshift, L):
    p = lws_amd.lws(fsize, fshift, L=L)
    return p.W_ai, p


@pytest.mark.parametrize("fsize,fshift,L,T,compat", [
    (64, 16, 5, 40, True), (64, 16, 5, 40, False), (1024, 256, 5, 70, True), (1024, 256, 5, 33, False),
    (2048, 512, 5, 20, True), (64, 32, 5, 50, False), (64, 8, 5, 30, False), (48, 16, 3, 25, False),
    (64, 16, 3, 25, True), (64, 16, 1, 25, True), (128, 32, 7, 25, True), (16, 4, 5, 30, True), (20, 5, 5, 30, True),
    (24, 6, 5, 30, True),
    # (Q-1)(L+1) > 32 tap groups: every round of the eight-lane sum is needed (round-2 advisor finding)
    (256, 64, 10, 25, True), (256, 64, 12, 20, True), (128, 32, 15, 20, True), (256, 64, 12, 20, False)])
def test_lds_kernel_equals_generic_engine_bit_for_bit(fsize, fshift, L, T, compat, monkeypatch):
    rng = np.random.default_rng(fsize + T + L)
    F = fsize // 2 + 1
    if L > F - 2:
        pytest.skip("L too large for F")
    W, p = weights(fsize, fshift, L)
    S = rng.standard_normal((3, T, F)) + 1j * rng.standard_normal((3, T, F))
    S[2] = np.abs(S[2])                                  # zero-phase magnitudes: the run_lws(abs(X)) case
    lds = _capi.Plan(F, W, nofuture_q4_compat=compat)
    gen = _capi.Plan(F, W, nofuture_q4_compat=compat, force_generic=True)
    for thr in ([0.0], [0.9, 0.4, 0.0]):
        monkeypatch.setenv("LWS_NOFUTURE_SERIAL_TAPS", "1")
        a = lds.nofuture(S, thr)
        name = lds.last_kernel()["name"]
        Q = W.shape[1]
        assert name == ("nofuture_lds_q4compat_fp32" if (compat and Q == 4) else "nofuture_lds_fp32"), name
        b = gen.nofuture(S, thr)
        assert gen.last_kernel()["name"] == "generic_fp32"
        assert np.array_equal(a, b), (fsize, fshift, L, T, compat, len(thr))
        # production variant: the same bins updated to the same magnitudes; values within rounding where rounding stays
        # rounding (the shipped Q4 addressing chains the bins of a frame and amplifies it: median only)
        monkeypatch.delenv("LWS_NOFUTURE_SERIAL_TAPS")
        c = lds.nofuture(S, thr)
        assert lds.last_kernel()["name"] == name
        assert np.abs(np.abs(c) - np.abs(b)).max() < 2e-6 * np.abs(S).max()
        S32 = S.astype(np.complex64)
        assert np.array_equal(c == S32, b == S32)                   # the same bins were left alone
        err = np.abs(c - b)
        if compat and Q == 4:
            # the shipped addressing is chaotic from frame to frame (two correct fp32 engines, or fp32 and fp64, decorrelate
            # within ~30 frames: rel-L2 1.0 between the fp32 generic engine and the fp64 oracle): values on the first frames
            assert np.median(err[:, :2]) < 2e-6 * np.mean(np.abs(S)) and np.quantile(err[:, 0], 0.99) < 1e-3 * np.mean(np.abs(S))
        else:
            # (rounding grows from frame to frame here too, only slower: 1e-7 on the first frames, 1e-3 after 40)
            assert np.median(err[:, :2]) < 2e-6 * np.mean(np.abs(S)) and np.median(err) < 1e-4 * np.mean(np.abs(S))
            assert np.linalg.norm(err) < 5e-2 * np.linalg.norm(b), np.linalg.norm(err) / np.linalg.norm(b)
    lds.close(); gen.close()


@pytest.mark.parametrize("fsize,fshift", [(16, 4), (20, 5), (24, 6), (28, 7), (64, 16)])
def test_compat_rounds_on_tiny_frames_against_the_oracle(oracle, fsize, fshift):
    """F <= 3L - 1: a bin of the upper range writes a Hermitian image that later bins of the same frame read through the
    flat offset; the rounds must keep the sequential order there (fp64 generic engine vs the oracle's sequential loop)."""
    rng = np.random.default_rng(fsize)
    F = fsize // 2 + 1
    L = min(5, F - 2)
    W, p = weights(fsize, fshift, L)
    S = rng.standard_normal((30, F)) + 1j * rng.standard_normal((30, F))
    g64 = _capi.Plan(F, W, precision="fp64")
    for thr in ([0.0], [0.5, 0.0]):
        ref = oracle.nofuture_lws(S, W, thr, compat=True)
        out = g64.nofuture(S, thr)
        # (an order violation shows up at 1e-2; what is left is re-association amplified by near-cancelling sums)
        assert np.abs(out - ref).max() < 1e-6, (fsize, np.abs(out - ref).max())
    g64.close()
    # (fp32 is pinned through bit-identity with the fp32 generic engine, above: the shipped addressing chains the bins of a
    # frame, and rounding is amplified along the chain until single fp32 values say little -- tests/test_gpu_parity.py)


@pytest.mark.parametrize("fsize,fshift,T", [(4096, 1024, 30), (3000, 750, 25), (4096, 2048, 20)])
def test_frames_of_4096_points(fsize, fshift, T, monkeypatch):
    """Frames of up to 3072 columns (4096-point STFTs) run on the LDS kernel's eight-lanes-per-bin variant (1024 threads: three
    columns of the prefetched frame each); the one-lane verification variant has 512 and sends them to the generic engine.
    Same magnitudes, same bins left alone, the generic engine's values on the first frames (see above for the later ones)."""
    rng = np.random.default_rng(T)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, mode="music")
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    thr = [0.3, 0.0]
    lds = _capi.Plan(F, p.W, p.W_ai, p.W_af)
    gen = _capi.Plan(F, p.W, p.W_ai, p.W_af, force_generic=True)
    a, b = lds.nofuture(S, thr, wsel=1), gen.nofuture(S, thr, wsel=1)
    assert lds.last_kernel()["name"].startswith("nofuture_lds") and gen.last_kernel()["name"] == "generic_fp32"
    assert np.abs(np.abs(a) - np.abs(b)).max() < 2e-6 * np.abs(S).max()
    S32 = S.astype(np.complex64)
    assert np.array_equal(a == S32, b == S32)
    err = np.abs(a - b)
    assert np.median(err[:, :2]) < 2e-6 * np.mean(np.abs(S)) and np.quantile(err[:, 0], 0.99) < 1e-3 * np.mean(np.abs(S))
    monkeypatch.setenv("LWS_NOFUTURE_SERIAL_TAPS", "1")
    ser = _capi.Plan(F, p.W, p.W_ai, p.W_af)
    c = ser.nofuture(S, thr, wsel=1)
    # (one lane per bin: 512 threads x 3 columns -- a frame of more than 1536 columns goes to the generic engine; the same bits either way)
    assert ser.last_kernel()["name"] == ("generic_fp32" if F + 10 > 1536 else "nofuture_lds_q4compat_fp32") and np.array_equal(c, b)
    lds.close(); gen.close(); ser.close()


def test_config3_nofuture_stage_time_and_values(monkeypatch):
    """256 x 500 x 513 (BASELINE config 3's first stage): one sweep; the verification variant gives the generic engine's bits
    on a sample, the production variant the same magnitudes."""
    import torch
    rng = np.random.default_rng(1)
    p = lws_amd.lws(1024, 256, mode="music")
    B, T, F = 64, 500, 513
    M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex64)
    t = torch.from_numpy(M).cuda()
    thr = lws_amd.get_thresholds(1, 1, 0.1, 1)
    stream = torch.cuda.current_stream().cuda_stream
    p.plan().nofuture_dev(t.data_ptr(), B, T, thr, wsel=1, stream=stream)
    info = p.plan().last_kernel()
    assert info["name"] == "nofuture_lds_q4compat_fp32" and info["ms"] < 5.6, info   # (the generic engine needs 17 ms, one lane per bin 6.2)
    pg = lws_amd.lws(1024, 256, mode="music", force_generic=True)
    t2 = torch.from_numpy(M[:4]).cuda()
    pg.plan().nofuture_dev(t2.data_ptr(), 4, T, thr, wsel=1, stream=stream)
    torch.cuda.synchronize()
    assert float((t[:4].abs() - t2.abs()).abs().max()) < 2e-6 * float(np.abs(M).max())
    monkeypatch.setenv("LWS_NOFUTURE_SERIAL_TAPS", "1")
    t3 = torch.from_numpy(M[:4]).cuda()
    p.plan().nofuture_dev(t3.data_ptr(), 4, T, thr, wsel=1, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(t3, t2)
