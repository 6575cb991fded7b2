"""GPU (-m gpu): the host-array entry points (numpy complex128 in / out; lws.pyx:209-258) of an fp32 plan -- the chunked, pinned,
overlapped pipeline of lws_capi.hip: run_host_pipelined.  Cutting the batch into chunks must change nothing; a bin no sweep
updated comes back as the caller's complex128 value bit for bit; the values are those of the *_dev entry points."""
import ctypes

import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def _data(B, T, F, seed, magnitudes):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))
    return np.abs(S).astype(np.complex128) if magnitudes else S


@pytest.mark.parametrize("B,T,fsize,fshift,magnitudes,thr", [
    (12, 200, 1024, 256, True, "default"), (12, 200, 1024, 256, False, "alpha1"), (12, 200, 1024, 256, True, "dense"),
    (7, 90, 1000, 250, True, "alpha1"), (5, 33, 2048, 512, False, "dense"), (1, 40, 64, 16, True, "alpha1")])
def test_batch_chunks_change_nothing(B, T, fsize, fshift, magnitudes, thr, monkeypatch):
    import torch
    F = fsize // 2 + 1
    S = _data(B, T, F, B * T, magnitudes)
    thr = {"default": lws_amd.get_thresholds(100, 100, 0.1, 1), "alpha1": lws_amd.get_thresholds(30, 1, 0.1, 1), "dense": np.zeros(20)}[thr]
    plan = lws_amd.lws(fsize, fshift).plan()
    one = plan.batch(S, thr)                                        # (one chunk at these sizes)
    for bins in (str(2 * T * F + 5), str(3 * T * F - 1), str(T * F)):
        monkeypatch.setenv("LWS_HOST_CHUNK_BINS", bins)
        monkeypatch.setenv("LWS_HOST_CHUNK_EXACT", "1")             # (else the chunks are rounded to divisors of the CU count)
        assert np.array_equal(plan.batch(S, thr), one), bins
        monkeypatch.delenv("LWS_HOST_CHUNK_EXACT")
        assert np.array_equal(plan.batch(S, thr), one), bins
    monkeypatch.setenv("LWS_HOST_TRACE", "1")                       # the diagnostic output must not disturb anything
    assert np.array_equal(plan.batch(S, thr), one)
    monkeypatch.delenv("LWS_HOST_TRACE")
    monkeypatch.delenv("LWS_HOST_CHUNK_BINS")
    # the device entry point on the rounded values
    d = torch.from_numpy(S.astype(np.complex64)).cuda()
    plan.batch_dev(d.data_ptr(), B, T, thr)
    torch.cuda.synchronize()
    assert np.abs(one - d.cpu().numpy()).max() < 1e-6 * np.abs(S).max()
    # the round-2 path (one complex128 copy each way around the kernels): same values for magnitude inputs; same bins untouched
    monkeypatch.setenv("LWS_HOST_MONOLITHIC", "1")
    mono = plan.batch(S, thr)
    if magnitudes:
        assert np.array_equal(mono, one)
    assert np.array_equal(one == S, mono == S) or not magnitudes
    # magnitudes are preserved to fp32 accuracy whatever happened to the phase
    assert np.abs(np.abs(one) - np.abs(S)).max() < 2e-6 * np.abs(S).max()


def test_pipeline_chunks_hold_whole_devices(monkeypatch):
    """A call with a one-workgroup-per-spectrogram stage (run_lws(mode='music'): no-future, online) is cut into chunks of a
    multiple of the CU count (a launch of 64 spectrograms takes as long as one of 256 there); same results as one chunk."""
    lib = _capi.load()
    n_cu = 256
    B, T, fsize, fshift = n_cu + 44, 24, 256, 64
    F = fsize // 2 + 1
    S = _data(B, T, F, 5, True)
    p = lws_amd.lws(fsize, fshift, mode="music", online_iterations=3, batch_iterations=10)
    monkeypatch.setenv("LWS_HOST_CHUNK_BINS", str(1 << 30))
    one = p.run_lws(S)
    monkeypatch.setenv("LWS_HOST_CHUNK_BINS", str(20 * T * F))      # 20 spectrograms' worth: becomes 256 + 44
    monkeypatch.setenv("LWS_HOST_TRACE", "1")
    two = p.run_lws(S)
    assert np.array_equal(one, two)
    monkeypatch.setenv("LWS_HOST_CHUNK_EXACT", "1")                 # 20 | 20 | ... as asked
    many = p.run_lws(S)
    assert np.array_equal(one, many)
    assert lib.lws_device_count() >= 1


def test_in_place_and_small_calls():
    """S_out == S_in through the C ABI (lws_hip.h allows it), and calls far below one chunk."""
    lib = _capi.load()
    p = lws_amd.lws(64, 16)
    plan = p.plan()
    S = _data(3, 17, 33, 9, False)
    thr = np.zeros(6)
    ref = plan.batch(S, thr)
    buf = S.copy()
    _capi.check(lib.lws_batch_lws(plan._h, 0, buf.ctypes.data, buf.ctypes.data, 3, 17, thr.ctypes.data, 6))
    assert np.array_equal(buf, ref)
    one = plan.batch(S[:1, :1], thr)
    assert one.shape == (1, 1, 33) and np.abs(np.abs(one) - np.abs(S[:1, :1])).max() < 1e-6 * np.abs(S).max()


@pytest.mark.parametrize("where", ["never", "first", "chunk2", "last"])
def test_real_valued_input_goes_up_as_four_bytes_per_bin(where, monkeypatch):
    """Magnitudes -- the documented usage run_lws(np.abs(X)), python/README.md:98-100 -- are uploaded as 4 bytes per bin and expanded
    on the device.  The decision is made chunk by chunk while narrowing: same bits as the complex upload (LWS_HOST_REAL=0) for
    real input, for complex input, and for input that turns complex in its first element / in a later chunk / in its last
    element; and a stream-ordering check rides along: a *_dev call on the same plan enqueued just before must not be overtaken."""
    import torch
    B, T, fsize, fshift = 9, 60, 512, 128
    F = fsize // 2 + 1
    S = _data(B, T, F, 77, True)
    if where == "first": S[0, 0, 0] += 0.25j
    if where == "chunk2": S[5, 7, 11] += 0.25j
    if where == "last": S[-1, -1, -1] += 0.25j
    thr = lws_amd.get_thresholds(25, 1, 0.1, 1)
    plan = lws_amd.lws(fsize, fshift).plan()
    monkeypatch.setenv("LWS_HOST_CHUNK_BINS", str(2 * T * F))
    monkeypatch.setenv("LWS_HOST_CHUNK_EXACT", "1")
    monkeypatch.setenv("LWS_HOST_REAL", "0")
    ref = plan.batch(S, thr)
    monkeypatch.delenv("LWS_HOST_REAL")
    # a device-resident call on a side stream right before the host call: both use the plan's scratch
    d = torch.from_numpy(S.astype(np.complex64)).cuda()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    plan.batch_dev(d.data_ptr(), B, T, thr, stream=side.cuda_stream)
    out = plan.batch(S, thr)
    side.synchronize()
    assert np.array_equal(out, ref), where
    assert np.abs(d.cpu().numpy() - ref).max() < 1e-6 * np.abs(S).max()      # ... and was not disturbed by it
    monkeypatch.setenv("LWS_HOST_HALF_FIRST", "0")                           # the half-size first chunk changes nothing either
    assert np.array_equal(plan.batch(S, thr), ref)
