"""Device STFT / iSTFT / consistency (lws_amd/csrc/lws_stft.hip) against the host restatement of lws.pyx:43-144
(lws_amd.stft / istft / get_consistency, which tests/test_host_helpers.py pins to the reference's goldens).
fp32 transforms: 2e-6 of the largest value; consistency within 0.01 dB."""
import numpy as np
import pytest
import torch

import lws_amd

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_device_transforms_against_the_reference_goldens():
    """Directly against tests/golden/helpers.npz (made by importing the reference, tests/golden/make_golden.py:70-81): stft /
    istft of lws.lws(64,16) with and without perfectrec, and get_consistency -- not through the host restatement."""
    g = load_golden("helpers.npz")
    p = lws_amd.lws(64, 16)
    x = g["x"]
    X = p.stft_dev(x).cpu().numpy()
    assert X.shape == g["stft_64_16"].shape
    assert np.abs(X - g["stft_64_16"]).max() < 3e-6 * np.abs(g["stft_64_16"]).max()
    y = p.istft_dev(g["stft_64_16"]).cpu().numpy()
    assert y.shape == g["istft_64_16"].shape and np.abs(y - g["istft_64_16"]).max() < 3e-6 * np.abs(g["istft_64_16"]).max()
    Xn = lws_amd.stft_dev(x, 64, 16, p.awin, perfectrec=False).cpu().numpy()                      # lws.pyx:43-90, module level
    assert Xn.shape == g["stft_np_64_16"].shape and np.abs(Xn - g["stft_np_64_16"]).max() < 3e-6 * np.abs(g["stft_np_64_16"]).max()
    yn = lws_amd.istft_dev(g["stft_np_64_16"], 16, p.swin, perfectrec=False).cpu().numpy()        # lws.pyx:93-137
    assert yn.shape == g["istft_np_64_16"].shape and np.abs(yn - g["istft_np_64_16"]).max() < 3e-6 * np.abs(g["istft_np_64_16"]).max()
    c = p.get_consistency_dev(np.abs(g["stft_64_16"]).astype(complex))
    assert abs(c - float(g["consistency_64_16"])) < 0.01


@pytest.mark.parametrize("fsize,fftsize,fshift", [(64, 128, 16), (48, 96, 16), (100, 128, 40), (400, 512, 160), (1000, 1024, 250), (1024, 4096, 256), (250, 1000, 100)])
@pytest.mark.parametrize("perfectrec", [True, False])
def test_transform_longer_than_the_frame(fsize, fftsize, fshift, perfectrec):
    """fftsize > fsize (lws.pyx:49-50,85: np.fft.fft(frame, n=fftsize) of the fsize windowed samples; the frame count and padding
    follow fsize): against the reference goldens of tests/golden/fftsize.npz where there is one, else against the host restatement
    (which tests/test_host_helpers.py pins to those goldens).  The inverse with fftsize != frame is a ValueError, as in the reference."""
    g = load_golden("fftsize.npz")
    k = "%d_%d_%d" % (fsize, fftsize, fshift)
    if f"awin_{k}" in g.files:
        X = lws_amd.stft_dev(g["x"], fsize, fshift, g[f"awin_{k}"], fftsize=fftsize, perfectrec=perfectrec).cpu().numpy()
        ref = g[f"stft_{k}_{int(perfectrec)}"]
        assert X.shape == ref.shape and np.abs(X - ref).max() < 3e-6 * np.abs(ref).max()
    rng = np.random.default_rng(fsize + fftsize)
    awin = np.sqrt(lws_amd.hann(fsize, symmetric=True, use_offset=False))
    n = 9 * fsize + 13
    x = rng.standard_normal((2, n))
    S = lws_amd.stft_dev(x, fsize, fshift, awin, fftsize=fftsize, perfectrec=perfectrec).cpu().numpy()
    ref = np.stack([lws_amd.stft(x[b], fsize, fshift, awin, fftsize=fftsize, perfectrec=perfectrec) for b in range(2)])
    assert S.shape == ref.shape == (2, ref.shape[1], fftsize // 2 + 1)
    assert np.abs(S - ref).max() < 3e-6 * np.abs(ref).max()
    swin = lws_amd.synthwin(awin, fshift)
    spec = rng.standard_normal((2, 11, fsize // 2 + 1)) + 1j * rng.standard_normal((2, 11, fsize // 2 + 1))
    with pytest.raises(ValueError):
        lws_amd.istft_dev(spec, fshift, swin, fftsize=fftsize)
    with pytest.raises(ValueError):
        lws_amd.stft_dev(x, fsize, fshift, awin, fftsize=fftsize + 1)
    with pytest.raises((ValueError, lws_amd.LwsHipError)):
        lws_amd.stft_dev(x, fsize, fshift, awin, fftsize=fsize - 2)           # a transform shorter than the frame


def test_class_with_fftsize_pads_its_windows():
    """class lws(fsize, fshift, fftsize=...) zero-pads its windows symmetrically (lws.pyx:396-411) and works on fftsize-point frames
    from there on: the device transforms against the reference golden."""
    import contextlib, io
    g = load_golden("fftsize.npz")
    with contextlib.redirect_stdout(io.StringIO()):
        p = lws_amd.lws(64, 16, fftsize=96)
    X = p.stft_dev(g["x"]).cpu().numpy()
    assert np.abs(X - g["cls_stft_64_96_16"]).max() < 3e-6 * np.abs(g["cls_stft_64_96_16"]).max()
    y = p.istft_dev(g["cls_stft_64_96_16"]).cpu().numpy()
    assert np.abs(y - g["cls_istft_64_96_16"]).max() < 3e-6 * np.abs(g["cls_istft_64_96_16"]).max()


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
