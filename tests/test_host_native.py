"""CPU: the native (C++) window / weight / schedule construction of liblws_hip.so (include/lws_hip.h, lws_host.cpp)
against the Python restatement -- which tests/test_host_helpers.py pins to goldens made by the reference -- and
against those goldens directly.  No GPU needed: these entry points do no device work."""
import ctypes as C

import numpy as np
import pytest

import lws_amd
from lws_amd import _capi
from conftest import load_golden


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    return _capi.load()


@pytest.mark.parametrize("n", [8, 33, 64, 1024])
@pytest.mark.parametrize("sym,off", [(1, 0), (0, 0), (0, 1)])
def test_hann(lib, n, sym, off):
    out = np.empty(n)
    assert lib.lws_hann(n, sym, off, _p(out)) == 0
    assert np.abs(out - lws_amd.hann(n, symmetric=bool(sym), use_offset=bool(off))).max() < 1e-15


@pytest.mark.parametrize("fsize,fshift", [(64, 16), (64, 32), (64, 8), (48, 16), (32, 12), (1024, 256)])
def test_synthwin_weights_and_asymmetric_windows(lib, fsize, fshift):
    awin = np.sqrt(lws_amd.hann(fsize))
    swin = np.empty(fsize)
    assert lib.lws_synthwin(_p(awin), fsize, fshift, None, _p(swin)) == 0
    ref_swin = lws_amd.synthwin(awin, fshift)
    assert np.abs(swin - ref_swin).max() < 1e-14
    rng = np.random.default_rng(fsize)
    other = np.abs(rng.standard_normal(fsize)) + 0.1
    assert lib.lws_synthwin(_p(awin), fsize, fshift, _p(other), _p(swin)) == 0
    assert np.abs(swin - lws_amd.synthwin(awin, fshift, swin=other)).max() < 1e-13
    for summarized in (1, 0):
        for L in (1, 5):
            qp, q = C.c_int(), C.c_int()
            assert lib.lws_weights_shape(fsize, fshift, summarized, C.byref(qp), C.byref(q)) == 0
            ref = lws_amd.create_weights(awin, ref_swin, fshift, L, use_summarized_weights=bool(summarized))
            assert ref.shape == (qp.value, q.value, L + 1)
            W = np.empty(ref.shape, dtype=np.complex128)
            assert lib.lws_create_weights(_p(awin), _p(ref_swin), fsize, fshift, L, summarized, _p(W)) == 0
            assert np.abs(W - ref).max() < 1e-14
    ai, af = np.empty(fsize), np.empty(fsize)
    assert lib.lws_build_asymmetric_windows(_p(awin * ref_swin), fsize, fshift, _p(ai), _p(af)) == 0
    rai, raf = lws_amd.build_asymmetric_windows(awin * ref_swin, fshift)
    assert np.abs(ai - rai).max() < 1e-14 and np.abs(af - raf).max() < 1e-14


@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8", "48_16"])
def test_weights_match_the_reference_goldens(lib, tag):
    """helpers.npz holds W / W_ai / W_af of lws.lws(fsize, fshift, mode='music') made by the reference itself."""
    h = load_golden("helpers.npz")
    fsize, fshift = [int(v) for v in tag.split("_")]
    hw = np.empty(fsize)
    lib.lws_hann(fsize, 1, 0, _p(hw))
    awin = np.sqrt(hw)
    sw = np.empty(fsize)
    lib.lws_synthwin(_p(awin), fsize, fshift, None, _p(sw))
    awin = np.sqrt(awin * sw)                               # lws.pyx:386-388
    lib.lws_synthwin(_p(awin), fsize, fshift, None, _p(sw))
    ai, af = np.empty(fsize), np.empty(fsize)
    lib.lws_build_asymmetric_windows(_p(awin * sw), fsize, fshift, _p(ai), _p(af))
    for name, win in (("W", awin), ("W_ai", ai), ("W_af", af)):
        ref = h[f"{name}_{tag}"]
        W = np.empty(ref.shape, dtype=np.complex128)
        assert lib.lws_create_weights(_p(np.ascontiguousarray(win)), _p(sw), fsize, fshift, ref.shape[2] - 1, 1, _p(W)) == 0
        assert np.abs(W - ref).max() < 1e-13, name


def test_get_thresholds_and_errors(lib):
    out = np.empty(100)
    assert lib.lws_get_thresholds(100, 100.0, 0.1, 1.0, _p(out)) == 0
    assert np.abs(out - lws_amd.get_thresholds(100, 100, 0.1, 1)).max() < 1e-12
    assert lib.lws_get_thresholds(7, 2.0, 0.4, 1.5, _p(out)) == 0
    assert np.abs(out[:7] - lws_amd.get_thresholds(7, 2.0, 0.4, 1.5)).max() < 1e-14
    assert lib.lws_get_thresholds(0, 1.0, 1.0, 1.0, None) == 0
    w = np.zeros(16)
    assert lib.lws_synthwin(_p(w), 16, 4, None, _p(w.copy())) == _capi.LWS_ERR_INVALID
    assert b"normalizer" in lib.lws_last_error()


@pytest.mark.gpu
def test_plan_from_windows_equals_the_python_built_plan():
    lib = _capi.load()
    rng = np.random.default_rng(0)
    for fsize, fshift in ((64, 16), (128, 64), (1024, 256)):
        p = lws_amd.lws(fsize, fshift, mode="music")
        F = fsize // 2 + 1
        S = rng.standard_normal((2, 20, F)) + 1j * rng.standard_normal((2, 20, F))
        thr = np.array([0.5, 0.2, 0.0])
        ref = p.plan().batch(S, thr)
        h = C.c_void_p()
        aw, sw = np.empty(fsize), np.empty(fsize)
        assert lib.lws_plan_create_from_windows(C.byref(h), 0, None, None, fsize, fshift, 5, 1, 0, _p(aw), _p(sw)) == 0
        assert np.abs(aw - p.awin).max() < 1e-14 and np.abs(sw - p.swin).max() < 1e-14
        out = np.empty_like(S)
        assert lib.lws_batch_lws(h, 0, _p(S), _p(out), 2, 20, _p(thr), 3) == 0
        assert np.abs(out - ref).max() < 2e-5 * np.abs(S).max()
        on_ref = p.plan().online(S, thr, 3, fsize / fshift)
        assert lib.lws_online_lws(h, _p(S), _p(out), 2, 20, _p(thr), 3, 3, C.c_double(fsize / fshift)) == 0
        assert np.linalg.norm(out - on_ref) < 1e-3 * np.linalg.norm(on_ref)
        lib.lws_plan_destroy(h)
