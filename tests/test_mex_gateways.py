"""The MATLAB gateways of matlab/*.cpp, compiled against the test-only mx stub of
tests/mex_stub/ (no MATLAB or Octave exists here or on the GPU box) and driven with numpy buffers.

CPU: they compile, link against the C-ABI library and reproduce the reference gateways' argument checks.
GPU: with LWS_MEX_FP64=1 they reproduce the reference wrapper goldens (the Python wrappers and the mex gateways run
the same kernels on the same prepared data; they differ only in the summation order of mean|S|, a last-ulp effect)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, have_gpu, load_golden

GATEWAYS = ("batch_lws", "nofuture_lws", "online_lws")


@pytest.fixture(scope="module")
def gateways(tmp_path_factory):
    import lws_amd._capi as capi
    capi.load()  # builds nothing; makes sure the library is there (and torch's HIP runtime is initialised first)
    out = tmp_path_factory.mktemp("mex")
    libs = {}
    for g in GATEWAYS:
        so = str(out / f"{g}.so")
        subprocess.check_call(
            ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "tests", "mex_stub"),
             "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", g + ".cpp"),
             os.path.join(ROOT, "tests", "mex_stub", "mex_stub.cpp"), "-L", os.path.join(ROOT, "lws_amd"), "-llws_hip",
             "-Wl,-rpath," + os.path.join(ROOT, "lws_amd"), "-o", so])
        lib = C.CDLL(so)
        lib.mexstub_call.restype = C.c_int
        libs[g] = lib
    yield libs
    for lib in libs.values():
        lib.mexstub_exit()


def call(lib, *arrays):
    """arrays in MATLAB orientation (column-major dims); returns (output or None, log)."""
    n = len(arrays)
    keep, re, im, nd, dims = [], [], [], [], []
    for a in arrays:
        a = np.asarray(a)
        shape = list(a.shape) if a.ndim >= 2 else [1, a.size]
        f = np.asfortranarray(a.reshape(shape))
        r = np.ascontiguousarray(f.real.ravel(order="F"), dtype=np.float64)
        i = np.ascontiguousarray(f.imag.ravel(order="F"), dtype=np.float64) if np.iscomplexobj(a) else None
        keep += [r, i]
        re.append(r.ctypes.data)
        im.append(i.ctypes.data if i is not None else None)
        nd.append(len(shape))
        dims += shape + [1] * (3 - len(shape))
    cap = max(int(np.asarray(arrays[0]).size), 1)
    o_re, o_im = np.zeros(cap), np.zeros(cap)
    o_dims = (C.c_long * 3)()
    log = C.create_string_buffer(4096)
    got = lib.mexstub_call(n, (C.c_void_p * n)(*re), (C.c_void_p * n)(*im), (C.c_int * n)(*nd),
                           (C.c_long * (3 * n))(*dims), o_re.ctypes.data_as(C.c_void_p),
                           o_im.ctypes.data_as(C.c_void_p), o_dims, C.c_long(cap), log, 4096)
    if not got:
        return None, log.value.decode()
    shape = [int(d) for d in o_dims]
    while len(shape) > 2 and shape[-1] == 1:
        shape.pop()
    return (o_re + 1j * o_im).reshape(shape, order="F"), log.value.decode()


def matlab_weights(W):
    """W[Q'][Q][L+1] (Python) -> (L+1) x Q x Q' (create_weights.m): the same memory, MATLAB dims."""
    return np.transpose(W, (2, 1, 0))


def test_gateways_check_arguments_like_the_reference(gateways):
    h = load_golden("helpers.npz")
    W = matlab_weights(h["W_64_16"])
    thr = np.array([0.1, 0.05])
    out, log = call(gateways["batch_lws"], np.ones((33, 6)), W)
    assert out is None and "not enought inputs" in log                       # batch_lws.cpp:27-30
    out, log = call(gateways["batch_lws"], np.ones((32, 6)), W, thr)
    assert out is None and "non-negative frequencies" in log                 # batch_lws.cpp:73-76
    out, log = call(gateways["nofuture_lws"], np.ones((33, 6)), W, np.ones((2, 2)))
    assert out is None and "1-D list of phase update thresholds" in log      # nofuture_lws.cpp
    out, log = call(gateways["batch_lws"], np.ones((33, 6)), W[:, :, 0], thr)
    assert out is None and "3-dimensional" in log                            # batch_lws.cpp:45-47
    out, log = call(gateways["online_lws"], np.ones((33, 6)), W, W, W, thr, np.array([1.0, 2.0]))
    assert out is None and "look-ahead" in log                               # online_lws.cpp:66-69
    if not have_gpu():  # no CPU fallback: the plan cannot be created, the gateway says so and returns nothing
        out, log = call(gateways["batch_lws"], np.ones((33, 6)), W, thr)
        assert out is None and log.startswith("lws:")


REFERENCE = os.environ.get("LWS_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "matlab", "batch_lws.cpp")),
                    reason="the reference tree is only mounted in the build container")
def test_reference_gateways_compile_and_link_against_the_compat_header(tmp_path):
    """the reference's OWN mex gateways (matlab/{batch,nofuture,online}_lws.cpp, compiled where they lie -- nothing is
    copied into the repository and the objects stay in pytest's tmp dir, they do not travel) build unchanged against the product's
    replacement for lwslib.h (include/lwslib_compat.h, reached through a one-line lwslib.h on the include path) and the test-only mx
    stub, and every lwslib symbol they need is resolved by liblws_hip.so under the reference's C++ mangling."""
    import lws_amd._capi as capi
    capi.load()
    inc = tmp_path / "inc"
    inc.mkdir()
    (inc / "lwslib.h").write_text('#include "lwslib_compat.h"\n')
    want = {"batch_lws": ["_Z5LWSQ4PdS_S_S_PiS_iiid", "_Z5LWSQ2PdS_S_S_PiS_iiid", "_Z7LWSanyQPdS_S_S_PiS_iiiid", "_Z10ExtendSpecPdS_S_S_iiii"],
            "nofuture_lws": ["_Z14NoFuture_LWSQ4PdS_S_S_PiS_iiid", "_Z16NoFuture_LWSanyQPdS_S_S_PiS_iiiid"],
            "online_lws": ["_Z11TF_RTISI_LAPdS_S_S_S_S_S_S_PiS0_S0_S_iiiiiidiS_i", "_Z8CopySpecPdS_S_S_iiii"]}
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "lws_amd", "liblws_hip.so")], capture_output=True, text=True).stdout
    for g in GATEWAYS:
        so = str(tmp_path / f"ref_{g}.so")
        out = subprocess.run(
            ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-w", "-Wl,--no-undefined", "-I", str(inc), "-I", os.path.join(ROOT, "tests", "mex_stub"),
             "-I", os.path.join(ROOT, "include"), os.path.join(REFERENCE, "matlab", g + ".cpp"),
             os.path.join(ROOT, "tests", "mex_stub", "mex_stub.cpp"), "-L", os.path.join(ROOT, "lws_amd"), "-llws_hip",
             "-Wl,-rpath," + os.path.join(ROOT, "lws_amd"), "-o", so], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-3000:]
        undef = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
        for sym in want[g]:
            assert sym in undef, (g, sym)          # the gateway calls it ...
            assert sym in exported, sym            # ... and the MI355X library provides it
        lib = C.CDLL(so)                            # (RTLD_NOW: every symbol resolves)
        lib.mexstub_call.restype = C.c_int
        out_, log = call(lib, np.ones((33, 6)))     # the reference's own argument check, through the stub
        assert out_ is None and "not enought inputs" in log
        if not have_gpu():
            continue
        # on a box with a GPU and the reference tree (none today): the reference's gateway code on the MI355X engine
        g_, h = load_golden("wrappers.npz"), load_golden("helpers.npz")
        W, W_ai, W_af = (matlab_weights(h[f"{k}_64_16"]) for k in ("W", "W_ai", "W_af"))
        S, thr = g_["S_64_16"], g_["thr_64_16"]
        if g == "batch_lws":
            o, log = call(lib, S.T, W, thr)
            assert np.abs(o.T - g_["batch_64_16"]).max() < 1e-8
        elif g == "nofuture_lws":
            o, log = call(lib, S.T, W_ai, thr[:2])
            assert np.abs(o.T - g_["nofuture_64_16"]).max() < 1e-8
        else:
            o, log = call(lib, S.T, W, W_ai, W_af, thr[:3], np.array(3.0))
            assert np.abs(o.T - g_["online_64_16"]).max() < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8"])
def test_gateways_reproduce_wrapper_goldens(gateways, tag, monkeypatch):
    monkeypatch.setenv("LWS_MEX_FP64", "1")
    g, h = load_golden("wrappers.npz"), load_golden("helpers.npz")
    W, W_ai, W_af = (matlab_weights(h[f"{k}_{tag}"]) for k in ("W", "W_ai", "W_af"))
    S, thr = g[f"S_{tag}"], g[f"thr_{tag}"]
    out, log = call(gateways["batch_lws"], S.T, W, thr)
    assert out is not None, log
    assert np.abs(out.T - g[f"batch_{tag}"]).max() < 1e-8
    out, log = call(gateways["batch_lws"], np.abs(S).T, W, thr)              # real input (batch_lws.cpp:83-88)
    assert np.abs(out.T - g[f"batch_mag_{tag}"]).max() < 1e-8
    out, log = call(gateways["nofuture_lws"], S.T, W_ai, thr[:2])
    assert np.abs(out.T - g[f"nofuture_{tag}"]).max() < 1e-8
    out, log = call(gateways["online_lws"], S.T, W, W_ai, W_af, thr[:3], np.array(3.0))
    assert np.abs(out.T - g[f"online_{tag}"]).max() < 1e-8


@pytest.mark.gpu
def test_gateway_accepts_a_stack_and_runs_fp32(gateways, monkeypatch):
    monkeypatch.delenv("LWS_MEX_FP64", raising=False)
    g, h = load_golden("wrappers.npz"), load_golden("helpers.npz")
    W = matlab_weights(h["W_64_16"])
    S, thr = g["S_64_16"], g["thr_64_16"]
    stack = np.stack([S.T, np.conj(S.T) * 0.5], axis=2)                      # Nreal x T x 2
    out, log = call(gateways["batch_lws"], stack, W, thr)
    assert out is not None and out.shape == stack.shape, log
    ref = g["batch_64_16"]
    assert np.linalg.norm(out[:, :, 0].T - ref) / np.linalg.norm(ref) < 2e-3
    one, _ = call(gateways["batch_lws"], stack[:, :, 1], W, thr)
    assert np.array_equal(one, out[:, :, 1])                                 # spectrograms are independent
