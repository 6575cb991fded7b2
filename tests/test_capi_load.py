"""CPU: the C-ABI library builds, loads and exports every symbol include/lws_hip.h declares.
No compute is attempted without a GPU; argument errors that are detected before any HIP call are."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, have_gpu
import lws_amd
from lws_amd import _capi


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lws_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lws_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lws_hip.h but not exported"
    assert set(names) == set(_capi.EXPORTS)
    assert lib.lws_hip_version() >= 100


def test_no_torch_or_cxx_types_in_abi():
    text = open(os.path.join(ROOT, "include", "lws_hip.h")).read()
    assert 'extern "C"' in text
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for bad in ("torch", "at::", "std::", "hipStream_t", "float2"):
        assert bad not in code


def test_argument_errors_without_gpu():
    lib = _capi.load()
    h = ctypes.c_void_p()
    W = np.zeros((4, 4, 6), np.complex128)
    rc = lib.lws_plan_create(ctypes.byref(h), 0, 32, 5, 4, 4, W.ctypes.data, None, None, 0)  # even F
    assert rc == _capi.LWS_ERR_INVALID
    assert b"non-negative frequencies" in lib.lws_last_error()
    rc = lib.lws_plan_create(ctypes.byref(h), 0, 33, 5, 4, 7, W.ctypes.data, None, None, 0)  # bad Qp
    assert rc == _capi.LWS_ERR_INVALID
    assert lib.lws_batch_lws(None, 0, None, None, 1, 1, None, 0) == _capi.LWS_ERR_INVALID


@pytest.mark.skipif(have_gpu(), reason="only meaningful on a box without a GPU")
def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: with no device the engine raises instead of silently computing elsewhere."""
    p = lws_amd.lws(64, 16, batch_iterations=3, batch_alpha=1)
    with pytest.raises((lws_amd.LwsHipError, ValueError)):
        p.batch_lws(np.ones((6, 33)))


def test_product_code_never_touches_the_checker():
    """oracle/ is test infrastructure: nothing under lws_amd/ or include/ may reference it."""
    for top in ("lws_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                    src = open(os.path.join(dirpath, f)).read().lower()
                    assert "oracle" not in src and "lwso_" not in src, f"{f} references the checker"


def test_library_exports_the_lwslib_h_interface():
    """include/lwslib_compat.h: the reference's 16 native entry points, under the reference's own mangled names
    (the table oracle.RefLib binds the compiled reference with)."""
    from oracle.oracle import RefLib
    text = open(os.path.join(ROOT, "include", "lwslib_compat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"^(?:void|const char \*)\s*([A-Za-z_0-9]+)\s*\(", text, flags=re.M))
    assert declared - {"lwslib_compat_last_error"} == set(RefLib.SYMS)
    lib = _capi.load_raw()
    for name, sym in RefLib.SYMS.items():
        assert hasattr(lib, sym), f"{name} ({sym}) not exported"
    assert hasattr(lib, "_Z24lwslib_compat_last_errorv")
