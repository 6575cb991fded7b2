"""GPU (-m gpu): the reference's own usage snippet (python/README.md:92-102) under the reference's module name, and the
two bindings of the C ABI (the Cython shim and ctypes) side by side."""
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_readme_snippet_runs_unchanged():
    import lws                                   # the reference's module name
    x = np.random.default_rng(0).standard_normal(40000)
    lws_processor = lws.lws(512, 128, mode="speech")      # 512: window length; 128: window shift
    X = lws_processor.stft(x)                    # where x is a single-channel waveform
    X0 = np.abs(X)                               # Magnitude spectrogram
    c0 = lws_processor.get_consistency(X0)
    X1 = lws_processor.run_lws(X0)               # reconstruction from magnitude
    c1 = lws_processor.get_consistency(X1)
    assert X1.dtype == np.complex128 and X1.shape == X0.shape
    assert c1 > c0 + 5.0, (c0, c1)
    assert np.abs(np.abs(X1) - X0).max() < 2e-6 * X0.max()
    assert lws.__version__ == "1.2.8"
    # module-level functions of the reference surface
    W = lws.create_weights(lws_processor.awin, lws_processor.swin, 128, 5)
    Y = lws.batch_lws(X0, W, lws.get_thresholds(10, 1.0, 0.1, 1))
    assert Y.shape == X0.shape


def test_cython_and_ctypes_bindings_agree():
    """The default binding is the Cython shim (lws_amd/_cylws, built by make -C lws_amd/csrc); LWS_BINDING=ctypes
    selects the ctypes one.  Same library, same results."""
    from lws_amd import _capi
    _capi.load()
    assert _capi.BINDING == "cython", "the Cython shim was not built / not importable"
    code = ("import numpy as np, lws_amd; from lws_amd import _capi; _capi.load(); "
            "p = lws_amd.lws(64, 16, batch_iterations=5, batch_alpha=1.0); "
            "S = np.random.default_rng(1).standard_normal((9, 33)) + 0j; "
            "print(_capi.BINDING, repr(float(np.abs(p.batch_lws(S)).sum())), repr(complex(p.batch_lws(S)[3, 7])))")
    outs = []
    for binding in ("cython", "ctypes"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                           env=dict(__import__("os").environ, LWS_BINDING=binding))
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.split())
    assert outs[0][0] == "cython" and outs[1][0] == "ctypes"
    assert outs[0][1:] == outs[1][1:]


def test_generic_engine_fallback_warns_once():
    """A shape only the generic engine serves (L = 12: beyond the band engine's stencils) says so once per plan;
    plans that asked for it (force_generic, fp64) stay quiet."""
    import warnings
    import lws_amd
    rng = np.random.default_rng(0)
    p = lws_amd.lws(1000, 250, L=12, batch_iterations=3, batch_alpha=1.0)
    S = np.abs(rng.standard_normal((6, 501)) + 1j * rng.standard_normal((6, 501)))
    with pytest.warns(RuntimeWarning, match="generic engine"):
        p.batch_lws(S)
    assert p.plan().last_kernel()["name"].startswith("generic")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        p.batch_lws(S)                                            # second call of the same plan: no warning
        lws_amd.lws(1000, 250, L=9, batch_iterations=3, batch_alpha=1.0, precision="fp64").batch_lws(S)
        lws_amd.lws(1024, 256, batch_iterations=3, batch_alpha=1.0).batch_lws(np.abs(rng.standard_normal((6, 513))))


def test_a_generic_stage_inside_a_pipeline_warns_too(monkeypatch):
    """run_lws(mode='music') with L = 7: the batch stage has its systolic build (frames 16 steps apart), the online stage has no
    LDS kernel for that stencil: it runs on the team engine (no warning) -- and on the generic engine when the team engine is
    switched off: the last kernel's name does not say so, the warning does (lws_generic_stage)."""
    import warnings
    import lws_amd
    rng = np.random.default_rng(1)
    S = np.abs(rng.standard_normal((9, 513)) + 1j * rng.standard_normal((9, 513)))
    p = lws_amd.lws(1024, 256, L=7, mode="music", online_iterations=2, batch_iterations=3, batch_alpha=1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        p.run_lws(S)
    assert "_l7_" in p.plan().last_kernel()["name"]
    monkeypatch.setenv("LWS_NO_TEAM", "1")
    p = lws_amd.lws(1024, 256, L=7, mode="music", online_iterations=2, batch_iterations=3, batch_alpha=1.0)
    with pytest.warns(RuntimeWarning, match="online stage"):
        p.run_lws(S)
    assert "_l7_" in p.plan().last_kernel()["name"]
