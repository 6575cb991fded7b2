"""GPU (-m gpu): the BASELINE configurations at their full sizes, through size-independent properties -- magnitudes
preserved, untouched bins bit-identical, the consistency the sweeps reach, results independent of where in the batch
a spectrogram sits -- plus oracle comparisons on the pieces an fp64 CPU sweep finishes in seconds.  (Config 2 / 3
at B = 1 against the oracle: tests/test_gpu_parity.py; the config-5 crop: tests/test_gpu_fp16.py.)"""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


def rayleigh(torch, dev, B, T, F, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((B, T, F), dtype=torch.float32, device=dev)
    for b in range(B):
        re = torch.randn((T, F), generator=g, device=dev)
        im = torch.randn((T, F), generator=g, device=dev)
        out[b] = torch.sqrt(re * re + im * im)
    return out


def consistency_db(p, t, idx):
    import torch
    sel = t[idx].contiguous()
    return p.get_consistency_dev(sel)


def test_config4_shard_1024_spectrograms(oracle):
    """BASELINE config 4, one GPU's shard: 1024 spectrograms of 500 x 513, 100 sweeps (4 rounds of 256 workgroups)."""
    import torch
    dev = torch.device("cuda", 0)
    B, T, F = 1024, 500, 513
    p = lws_amd.lws(1024, 256)
    mags = rayleigh(torch, dev, B, T, F, 1)
    # spectrograms 0, 300, 777 and 1023 are copies of the same data: a result must not depend on the batch position
    # (which CU / which round of workgroups ran it)
    for b in (300, 777, 1023):
        mags[b] = mags[0]
    state = mags.to(torch.complex64)
    thr = lws_amd.get_thresholds(100, 100, 0.1, 1)
    stream = torch.cuda.current_stream().cuda_stream
    p.plan().batch_dev(state.data_ptr(), B, T, thr, stream=stream)
    info = p.plan().last_kernel()
    assert info["name"] == "systolic_q4_l5_hann" and info["launches"] == 1
    assert torch.isfinite(torch.view_as_real(state)).all()
    assert float(((state.abs() - mags).abs().max() / mags.max()).item()) < 1e-6
    for b in (300, 777, 1023):
        assert torch.equal(state[b], state[0])
    # bins below the smallest threshold x mean were never updated: bit-identical
    mean = mags.mean(dim=(1, 2), keepdim=True)
    quiet = mags < 0.999 * float(thr.min()) * mean
    assert quiet.any() and torch.equal(state.real[quiet], mags[quiet]) and not state.imag[quiet].any()
    c = consistency_db(p, state, [0, 511, 1023])
    assert (c > 10.0).all() and (c < 13.0).all(), c          # one spectrogram of this kind reaches 11.04 dB (config-2 golden)
    # and spectrogram 0 against the oracle, value by value (SURVEY 8c tolerances)
    M = mags[0].cpu().numpy().astype(np.float64)
    ref = oracle.batch_lws(M, p.W, thr)
    out = state[0].cpu().numpy().astype(np.complex128)
    d = np.abs(out - ref)
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 1e-3 and np.median(d) < 1e-6 * M.mean()


def test_north_star_shape_1024_frames(oracle):
    """256 spectrograms of 1024 x 513 (the literal north-star shape), dense schedule: quality + one oracle comparison on
    the default schedule."""
    import torch
    dev = torch.device("cuda", 0)
    B, T, F = 256, 1024, 513
    p = lws_amd.lws(1024, 256)
    mags = rayleigh(torch, dev, B, T, F, 2)
    state = mags.to(torch.complex64)
    stream = torch.cuda.current_stream().cuda_stream
    p.plan().batch_dev(state.data_ptr(), B, T, np.zeros(100), stream=stream)
    assert p.plan().last_kernel()["name"] == "systolic_q4_l5_hann"
    assert float(((state.abs() - mags).abs().max() / mags.max()).item()) < 1e-6
    c = consistency_db(p, state, [0, 255])
    assert (c > 10.5).all(), c
    M = mags[7].cpu().numpy().astype(np.float64)
    thr = lws_amd.get_thresholds(100, 100, 0.1, 1)
    one = mags[7:8].to(torch.complex64)
    p.plan().batch_dev(one.data_ptr(), 1, T, thr, stream=stream)
    ref = oracle.batch_lws(M, p.W, thr)
    out = one[0].cpu().numpy().astype(np.complex128)
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 1e-3 and np.median(np.abs(out - ref)) < 1e-6 * M.mean()


@pytest.mark.parametrize("fsize,fshift,B,T,kernel", [(512, 128, 512, 500, "systolic_half_q4_l5_hann"), (256, 64, 300, 1000, "systolic_quarter_q4_l5_hann"),
                                                    (1000, 250, 256, 500, "systolic_q4_l5_hann")])
def test_config2_volume_at_other_frame_sizes(oracle, fsize, fshift, B, T, kernel, monkeypatch):
    """Round 3's new fast paths at config 2's volume: 257- and 129-bin frames (BASELINE config 1's frame size: two / four sweep
    slots per wave) and 501-bin frames (frame ends inside a block).  100 default-schedule sweeps from zero phase: batch position
    must not matter, untouched bins stay bit-identical, consistency as for 513 bins, spectrogram 0 against the oracle value by
    value; and the short-frame builds against the one-slot-per-wave build, bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift)
    mags = rayleigh(torch, dev, B, T, F, 3)
    for b in (B // 3, B - 1):
        mags[b] = mags[0]
    state = mags.to(torch.complex64)
    thr = lws_amd.get_thresholds(100, 100, 0.1, 1)
    stream = torch.cuda.current_stream().cuda_stream
    p.plan().batch_dev(state.data_ptr(), B, T, thr, stream=stream)
    info = p.plan().last_kernel()
    assert info["name"] == kernel and info["launches"] in (1, 2), info   # (2: a partial last round of workgroups shares the chip)
    assert torch.isfinite(torch.view_as_real(state)).all()
    assert float(((state.abs() - mags).abs().max() / mags.max()).item()) < 1e-6
    for b in (B // 3, B - 1):
        assert torch.equal(state[b], state[0])
    mean = mags.mean(dim=(1, 2), keepdim=True)
    quiet = mags < 0.999 * float(thr.min()) * mean
    assert quiet.any() and torch.equal(state.real[quiet], mags[quiet]) and not state.imag[quiet].any()
    c = consistency_db(p, state, [0, B // 2, B - 1])
    assert (c > 9.5).all() and (c < 13.5).all(), c
    M = mags[0].cpu().numpy().astype(np.float64)
    ref = oracle.batch_lws(M, p.W, thr)
    out = state[0].cpu().numpy().astype(np.complex128)
    assert np.linalg.norm(out - ref) / np.linalg.norm(ref) < 1e-3 and np.median(np.abs(out - ref)) < 1e-6 * M.mean()
    if F <= 257:
        monkeypatch.setenv("LWS_SYSTOLIC_NO_SHORT", "1")
        narrow = _capi.Plan(F, p.W)
        monkeypatch.delenv("LWS_SYSTOLIC_NO_SHORT")
        again = mags[:64].to(torch.complex64)
        narrow.batch_dev(again.data_ptr(), 64, T, thr, stream=stream)
        assert narrow.last_kernel()["name"] == "systolic_q4_l5_hann"
        assert torch.equal(again, state[:64])
        narrow.close()


@pytest.mark.parametrize("storage", ["fp32", "fp16"])
def test_config5_full_size(storage):
    """BASELINE config 5 at full size: 64 clips x 56 250 frames x 1025 bins, lws(2048, 512), 200 sweeps of the default
    schedule, fp32 and fp16-complex storage.  ~60 GB resident; several workgroups per clip hand rows over through HBM."""
    import torch
    dev = torch.device("cuda", 0)
    B, T, F = 64, 56250, 1025
    p = lws_amd.lws(2048, 512, storage=storage)
    mags = rayleigh(torch, dev, B, T, F, 3)
    mags[63] = mags[0]
    state = torch.empty((B, T, F), dtype=torch.complex64, device=dev)
    state.copy_(mags)
    thr = lws_amd.get_thresholds(200, 100, 0.1, 1)
    stream = torch.cuda.current_stream().cuda_stream
    p.plan().batch_dev(state.data_ptr(), B, T, thr, stream=stream)
    info = p.plan().last_kernel()
    assert info["name"] == "systolic_wide_q4_l5_hann" + ("_f16" if storage == "fp16" else ""), info
    assert "timed out" not in info["name"]
    for b0 in range(0, B, 8):                       # in slices: no batch-sized temporaries
        sl = slice(b0, b0 + 8)
        assert torch.isfinite(torch.view_as_real(state[sl])).all()
        assert float(((state[sl].abs() - mags[sl]).abs().max() / mags[sl].max()).item()) < 2e-6
    assert torch.equal(state[63], state[0])
    c = consistency_db(p, state, [0, 31])
    assert (c > 10.5).all() and (c < 14.0).all(), c
    del state, mags
    torch.cuda.empty_cache()
