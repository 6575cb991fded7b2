"""CPU: host-side mirror of the reference's helper functions (lws_amd/lws.py) against goldens
generated from the reference (lws.pyx:10-206), and the Python-level argument handling."""
import numpy as np
import pytest

import lws_amd
from conftest import load_golden

CFGS = ["64_16", "64_32", "64_8", "48_16", "64_24"]


@pytest.mark.parametrize("tag", CFGS)
def test_windows_and_weights(tag, capsys):
    g = load_golden("helpers.npz")
    fsize, fshift = [int(v) for v in tag.split("_")]
    p = lws_amd.lws(fsize, fshift)
    assert np.allclose(p.awin, g[f"awin_{tag}"], rtol=0, atol=1e-15)
    assert np.allclose(p.swin, g[f"swin_{tag}"], rtol=0, atol=1e-14)
    for name in ("W", "W_ai", "W_af"):
        assert getattr(p, name).shape == g[f"{name}_{tag}"].shape
        assert np.abs(getattr(p, name) - g[f"{name}_{tag}"]).max() < 1e-14
        # the participation mask (|w| > 1e-12, lws.pyx:231-232) must agree exactly
        assert np.array_equal(np.abs(getattr(p, name)) > 1e-12, np.abs(g[f"{name}_{tag}"]) > 1e-12)
    assert np.abs(p.win_ai - g[f"win_ai_{tag}"]).max() < 1e-14
    assert np.abs(p.win_af - g[f"win_af_{tag}"]).max() < 1e-14
    Wg = lws_amd.create_weights(p.awin, p.swin, fshift, 3, use_summarized_weights=False)
    assert Wg.shape == g[f"Wgen_{tag}"].shape == (fsize, p.W.shape[1], 4)
    assert np.abs(Wg - g[f"Wgen_{tag}"]).max() < 1e-14


def test_hann_thresholds_extspec():
    g = load_golden("helpers.npz")
    assert np.abs(lws_amd.hann(16) - g["hann_sym_16"]).max() < 1e-15
    assert np.abs(lws_amd.hann(16, symmetric=False) - g["hann_asym_16"]).max() < 1e-15
    assert np.abs(lws_amd.hann(16, symmetric=False, use_offset=True) - g["hann_asym_off_16"]).max() < 1e-15
    assert np.allclose(lws_amd.get_thresholds(100, 100, 0.1, 1), g["thr_100"], rtol=1e-15, atol=0)
    assert np.allclose(lws_amd.get_thresholds(7, 2.0, 0.3, 1.5), g["thr_gamma"], rtol=1e-15, atol=0)
    assert np.array_equal(lws_amd.extspec(g["ext_in"], 2, 3), g["ext_L2_Q3"])


def test_stft_istft_consistency():
    g = load_golden("helpers.npz")
    p = lws_amd.lws(64, 16)
    X = p.stft(g["x"])
    assert X.shape == g["stft_64_16"].shape and X.dtype == np.complex128
    assert np.abs(X - g["stft_64_16"]).max() < 1e-12
    assert np.abs(p.istft(g["stft_64_16"]) - g["istft_64_16"]).max() < 1e-12
    assert np.abs(p.istft(X)[:700] - g["x"]).max() < 1e-12  # perfect reconstruction (post-padded to a frame multiple)
    Xn = lws_amd.stft(g["x"], 64, 16, p.awin, perfectrec=False)
    assert np.abs(Xn - g["stft_np_64_16"]).max() < 1e-12
    assert np.abs(lws_amd.istft(Xn, 16, p.swin, perfectrec=False) - g["istft_np_64_16"]).max() < 1e-12
    c = p.get_consistency(np.abs(g["stft_64_16"]).astype(complex))
    assert abs(c - float(g["consistency_64_16"])) < 1e-9


def test_error_behaviour_matches_reference():
    p = lws_amd.lws(64, 16)
    with pytest.raises(ValueError, match="single channel"):
        lws_amd.stft(np.zeros((2, 100)), 64, 16, p.awin)
    with pytest.raises(ValueError, match="Odd ffts"):
        lws_amd.stft(np.zeros(100), 64, 16, p.awin, fftsize=65)
    with pytest.raises(ValueError, match="non-negative frequencies"):
        lws_amd.istft(np.zeros((4, 32), complex), 16, p.swin)
    with pytest.raises(ValueError, match="normalizer"):
        lws_amd.synthwin(np.zeros(16), 4)
    # even number of bins: ValueError before anything touches the GPU (lws.pyx:223-224)
    for fn in (lws_amd.batch_lws, lws_amd.nofuture_lws):
        with pytest.raises(ValueError, match="non-negative frequencies"):
            fn(np.ones((8, 32), complex), p.W, [0.0])
    with pytest.raises(ValueError, match="non-negative frequencies"):
        lws_amd.online_lws(np.ones((8, 32), complex), p.W, p.W_ai, p.W_af, np.array([0.0]), 3, 16)
    with pytest.raises(ValueError):
        p.batch_lws(np.ones((8, 32), complex))


def test_zero_iterations_returns_cast_input():
    p = lws_amd.lws(64, 16, batch_iterations=0)
    M = np.abs(np.random.default_rng(0).standard_normal((7, 33)))
    out = p.run_lws(M)  # all three stages have 0 iterations: no engine call at all
    assert out.dtype == np.complex128 and np.array_equal(out, M.astype(np.complex128))
    S = M.astype(np.complex128)
    assert lws_amd.batch_lws(S, p.W, []) is S  # lws.pyx:219-220 returns the (already complex128) input itself


def test_mode_presets_and_kwargs():
    p = lws_amd.lws(64, 16, mode="music")
    assert (p.nofuture_iterations, p.online_iterations, p.batch_iterations) == (1, 10, 100)
    p = lws_amd.lws(64, 16, mode="speech", nofuture_iterations=5, online_iterations=5)
    assert (p.nofuture_iterations, p.online_iterations) == (0, 0)
    assert p.Q == 4 and p.L == 5 and p.look_ahead == 3 and p.fsize == 64
    assert lws_amd.__version__ == "1.2.8"


def test_transform_longer_than_the_frame_goldens():
    """stft(..., fftsize > fsize) (lws.pyx:49-50,85), istft's ValueError for every fftsize != 2 (bins - 1) (lws.pyx:107-126), and
    class lws(fftsize=...), which pads its windows symmetrically instead (lws.pyx:396-411): tests/golden/fftsize.npz, made by
    importing the reference (make_golden.py fftsize)."""
    import contextlib, io
    g = load_golden("fftsize.npz")
    for k in ("64_128_16", "48_96_16", "100_128_40"):
        fsize, nfft, hop = (int(v) for v in k.split("_"))
        awin, swin = g[f"awin_{k}"], g[f"swin_{k}"]
        assert np.abs(lws_amd.synthwin(awin, hop) - swin).max() < 1e-14
        for pr in (False, True):
            X = lws_amd.stft(g["x"], fsize, hop, awin, fftsize=nfft, perfectrec=pr)
            ref = g[f"stft_{k}_{int(pr)}"]
            assert X.shape == ref.shape and np.abs(X - ref).max() < 1e-12
        assert int(g[f"istft_raises_{k}"]) == 1
        spec = np.ones((5, fsize // 2 + 1), complex)
        with pytest.raises(ValueError):
            lws_amd.istft(spec, hop, swin, fftsize=nfft)
        with pytest.raises(ValueError):
            lws_amd.istft(spec, hop, np.concatenate([swin, [0.0, 0.0]]))       # a window that is not the frame's length
        assert lws_amd.istft(spec, hop, swin).shape == (hop * 4 + fsize,)
    with contextlib.redirect_stdout(io.StringIO()) as out:
        p = lws_amd.lws(64, 16, fftsize=96)
    assert "Zero-padding symmetrically" in out.getvalue()
    assert p.fsize == 96 and np.abs(p.awin - g["cls_awin_64_96_16"]).max() < 1e-14 and np.abs(p.swin - g["cls_swin_64_96_16"]).max() < 1e-13
    assert np.abs(p.W - g["cls_W_64_96_16"]).max() < 1e-12
    X = p.stft(g["x"])
    assert np.abs(X - g["cls_stft_64_96_16"]).max() < 1e-12 and np.abs(p.istft(X) - g["cls_istft_64_96_16"]).max() < 1e-12
