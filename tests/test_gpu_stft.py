"""Device STFT / iSTFT / consistency (lws_amd/csrc/lws_stft.hip) against the host restatement of lws.pyx:43-144
(lws_amd.stft / istft / get_consistency, which tests/test_host_helpers.py pins to the reference's goldens).
fp32 transforms: 2e-6 of the largest value; consistency within 0.01 dB."""
import numpy as np
import pytest
import torch

import lws_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fsize,fshift", [(64, 16), (128, 64), (512, 128), (1024, 256), (2048, 512), (256, 96), (4096, 1024),
                                          # not powers of two: an odd factor (3, 125, 129, 5) times a power of two
                                          (48, 16), (1536, 384), (1000, 250), (1032, 258), (3840, 960)])
@pytest.mark.parametrize("perfectrec", [True, False])
def test_stft_istft_match_host(fsize, fshift, perfectrec):
    rng = np.random.default_rng(fsize + fshift)
    p = lws_amd.lws(fsize, fshift, perfectrec=perfectrec)
    n = 7 * fsize + 37
    x = rng.standard_normal((3, n))
    S = p.stft_dev(x).cpu().numpy()
    ref = np.stack([p.stft(x[b]) for b in range(3)])
    assert S.shape == ref.shape
    assert np.abs(S - ref).max() < 3e-6 * np.abs(ref).max()
    y = p.istft_dev(ref).cpu().numpy()
    yref = np.stack([p.istft(ref[b]) for b in range(3)])
    assert y.shape == yref.shape
    assert np.abs(y - yref).max() < 3e-6 * np.abs(yref).max()
    # single-signal forms
    assert np.abs(p.stft_dev(x[0]).cpu().numpy() - ref[0]).max() < 3e-6 * np.abs(ref).max()
    assert np.abs(p.istft_dev(ref[1]).cpu().numpy() - yref[1]).max() < 3e-6 * np.abs(yref).max()


@pytest.mark.parametrize("fsize,fshift", [(64, 16), (512, 128), (1024, 256), (1024, 512), (48, 16), (1536, 384)])
@pytest.mark.parametrize("perfectrec", [True, False])
def test_consistency_matches_host(fsize, fshift, perfectrec):
    rng = np.random.default_rng(3 * fsize + fshift)
    p = lws_amd.lws(fsize, fshift, perfectrec=perfectrec, batch_iterations=20, batch_alpha=1.0)
    F = fsize // 2 + 1
    X = p.stft(rng.standard_normal(12 * fsize))                # consistent by construction
    cases = [X, np.abs(X).astype(complex),                     # ... its magnitudes with zero phase
             rng.standard_normal((X.shape[0], F)) + 1j * rng.standard_normal((X.shape[0], F)),
             p.run_lws(np.abs(X))]
    for S in cases[1:]:
        assert abs(p.get_consistency_dev(S) - p.get_consistency(S)) < 0.01
    # a spectrogram that is consistent (up to the edge frames without perfectrec): where the host value is limited by
    # fp64 rounding, the fp32 transform bottoms out above 100 dB
    host = p.get_consistency(X)
    dev = p.get_consistency_dev(X)
    assert (abs(dev - host) < 0.01) if host < 60 else (dev > 100.0), (host, dev)
    stack = np.stack(cases[1:])
    db = p.get_consistency_dev(stack)
    assert db.shape == (3,)
    for i, S in enumerate(cases[1:]):
        assert abs(db[i] - p.get_consistency(S)) < 0.01
    # torch tensors on the device are used in place
    t = torch.from_numpy(stack.astype(np.complex64)).cuda()
    assert np.abs(p.get_consistency_dev(t) - db).max() < 1e-6


def test_unsupported_frame_sizes_raise():
    p = lws_amd.lws(8192, 2048)                  # beyond the LDS-resident transform (even sizes up to 4096)
    with pytest.raises(lws_amd.LwsHipError):
        p.get_consistency_dev(np.ones((5, 4097), complex))


def test_consistency_at_4096():
    rng = np.random.default_rng(5)
    p = lws_amd.lws(4096, 1024)
    X = p.stft(rng.standard_normal(10 * 4096))
    S = rng.standard_normal(X.shape) + 1j * rng.standard_normal(X.shape)
    assert abs(p.get_consistency_dev(S) - p.get_consistency(S)) < 0.01
    assert p.get_consistency_dev(X) > 100.0


@pytest.mark.gpu
def test_stream_copy_probe():
    """lws_stream_copy (the bench's measured-copy-peak kernel) copies exactly and rejects ragged sizes."""
    import torch
    from lws_amd import _capi
    lib = _capi.load()
    src = torch.randint(0, 255, (1 << 20,), dtype=torch.uint8, device="cuda")
    dst = torch.zeros_like(src)
    assert lib.lws_stream_copy(dst.data_ptr(), src.data_ptr(), src.numel(), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    assert lib.lws_stream_copy(dst.data_ptr(), src.data_ptr(), 24, None) == _capi.LWS_ERR_INVALID


def test_concurrent_streams_do_not_share_windows_or_scratch():
    """The transforms keep their windows and scratch in one context per device and are asynchronous on the caller's
    stream: calls with different windows / shapes enqueued on two streams without any host synchronisation in between
    must give what they give one at a time (the context serialises its users on the device)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    pa, pb = lws_amd.lws(1024, 256), lws_amd.lws(512, 128)
    xa = torch.from_numpy(rng.standard_normal((6, 60000)).astype(np.float32)).cuda()
    xb = torch.from_numpy(rng.standard_normal((3, 9000)).astype(np.float32)).cuda()
    ref_a, ref_b = pa.stft_dev(xa), pb.stft_dev(xb)
    ra, rb = pa.istft_dev(ref_a), pb.istft_dev(ref_b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(6):
        with torch.cuda.stream(s1):
            A = pa.stft_dev(xa)
            ia = pa.istft_dev(A)
        with torch.cuda.stream(s2):
            Bq = pb.stft_dev(xb)
            ib = pb.istft_dev(Bq)
        outs.append((A, ia, Bq, ib))
    torch.cuda.synchronize()
    for A, ia, Bq, ib in outs:
        assert torch.equal(A, ref_a) and torch.equal(Bq, ref_b)
        assert torch.equal(ia, ra) and torch.equal(ib, rb)
