"""The schedule of the band engine (lws_amd/csrc/lws_band.hip) on the CPU.

tests/band_emul.cpp compiles the very text the kernel's step is compiled from (lws_amd/csrc/lws_band_core.h, and the geometry /
table code of lws_band_host.h) with g++ and steps through it lane by lane -- same lane/frame mapping, ring rows and ages (asserted
in Lane::row_at), scatter order, image handling, frame period and lag as the kernel -- and must reproduce the oracle.  Runs
without a GPU; the kernel itself is tested in tests/test_gpu_band.py."""
import ctypes as C
import os
import subprocess
from math import gcd

import numpy as np
import pytest

import lws_amd

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libband_emul.so")


@pytest.fixture(scope="module")
def emul():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(HERE, "band_emul.cpp")
    deps = [src] + [os.path.join(HERE, "..", "lws_amd", "csrc", h) for h in ("lws_band_core.h", "lws_band_host.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        # -O1 keeps the asserts; no contraction: the emulation's fma_() calls are the only fused operations, as on the GPU
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", SO], check=True)
    lib = C.CDLL(SO)
    vp, ci = C.c_void_p, C.c_int
    lib.band_emul.argtypes = [vp, vp, ci, ci, vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, ci]
    lib.band_emul_helpers.argtypes = lib.band_emul.argtypes
    lib.band_emul_geometry.argtypes = [ci] * 7 + [vp]
    return lib


def run(lib, oracle, fs, hop, T, iters, L=5, NS=2, SKW=8, nls=64, LT=5, QT=8, fp32=0, seed=0, alpha=1.0, helpers=False):
    p = lws_amd.lws(fs, hop, L=L)
    W = np.ascontiguousarray(p.W)
    Qp, Q, L1 = W.shape
    F = fs // 2 + 1
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, alpha, 0.1, 1)
    ref = oracle.batch_lws(S, W, thr)
    # the twiddle of create_weights' tensors (lws.pyx:160-181): hop / frame in lowest terms; Q rows: one turn in Q bins
    g = gcd(fs, hop)
    Pt, s = (Q, 1) if Qp == Q else (fs // g, hop // g)
    ths = np.ascontiguousarray(thr * np.mean(np.abs(S)))
    out = np.empty_like(S)
    W0 = np.ascontiguousarray(W[0])
    rc = (lib.band_emul_helpers if helpers else lib.band_emul)(S.ctypes.data, out.ctypes.data, T, F, W0.ctypes.data, Q, L1 - 1, Pt, s, ths.ctypes.data, iters, NS, SKW, nls, LT, QT, fp32)
    assert rc == 0, rc
    return np.abs(out - ref).max() / np.abs(ref).max()


CASES = [
    # fs, hop, T, iters, kwargs
    (64, 16, 9, 4, {}),                                        # Q = 4, one block of frames
    (64, 16, 150, 3, dict(NS=3)),                              # lanes wrap into a second and third block
    (64, 32, 67, 4, {}),                                       # Q = 2
    (64, 8, 20, 3, {}),                                        # Q = 8
    (64, 4, 40, 3, dict(QT=16)),                               # Q = 16
    (48, 16, 30, 3, {}), (60, 12, 30, 3, {}), (60, 10, 30, 3, {}), (56, 8, 30, 3, {}),   # Q = 3, 5, 6, 7: twiddles from the table
    (64, 24, 30, 3, {}), (64, 20, 30, 3, {}), (400, 160, 20, 3, {}),                     # LWSfractionalQ: general tensors
    (64, 16, 30, 3, dict(L=3)),                                # narrower stencil on the LT = 5 build
    (64, 16, 30, 3, dict(L=7, LT=10, SKW=13)), (64, 16, 30, 3, dict(L=8, LT=10, SKW=12)), (64, 16, 30, 3, dict(L=10, LT=10, SKW=12)),
    (64, 8, 30, 2, dict(L=6, LT=10, SKW=12)),
    (64, 16, 70, 7, dict(NS=4, SKW=7)), (64, 16, 70, 5, dict(NS=1, SKW=9)),              # other skews, slot counts
    (1024, 256, 12, 3, {}),                                    # 513 bins on 64 lanes: a gap of 8 steps
    (1024, 256, 70, 2, dict(nls=128, SKW=9)),                  # two waves per slot
    (2048, 256, 10, 2, dict(nls=128)),                         # the shape the engine was built for
]


@pytest.mark.parametrize("fs,hop,T,iters,kw", CASES)
def test_emulated_schedule_reproduces_the_oracle(emul, oracle, fs, hop, T, iters, kw):
    err = run(emul, oracle, fs, hop, T, iters, **kw)
    # general tensors: create_weights' rows are the twiddle images of row 0 to ~1e-16 each (numpy's exp)
    assert err < (2e-12 if (fs % hop) else 2e-13), err


HELPER_CASES = [
    (64, 8, 20, 3, {}), (64, 8, 150, 5, dict(NS=3)), (64, 8, 114, 3, {}), (64, 8, 70, 3, dict(SKW=7, NS=1)),     # Q = 8: main r = 1..3, one helper r = 4..7
    (64, 4, 40, 3, dict(QT=16)),                                                                                  # Q = 16: three helpers
    (64, 16, 30, 3, dict(L=8, LT=10, QT=4, SKW=12)), (64, 16, 70, 4, dict(L=10, LT=10, QT=4, SKW=13, NS=4)),      # Q = 4, wide stencils
    (64, 16, 70, 4, dict(L=8, LT=8, QT=4, SKW=10, NS=4)), (64, 16, 70, 4, dict(L=6, LT=8, QT=4, SKW=11)),         # ... on the LT = 8 build
    (2048, 256, 10, 2, dict(nls=128)),
]


@pytest.mark.parametrize("fs,hop,T,iters,kw", HELPER_CASES)
def test_helper_waves_reproduce_the_oracle(emul, oracle, fs, hop, T, iters, kw):
    """The exact builds' slots are a main wave and helper waves a step ahead of it (lws_band_core.h: Lane, Split): the frame offsets
    shared out, the helpers' partial sums handed over through a mailbox two cells deep.  Same schedule on the CPU, ring ages asserted."""
    assert run(emul, oracle, fs, hop, T, iters, helpers=True, **kw) < 2e-13


def test_fp32_arithmetic(emul, oracle):
    assert run(emul, oracle, 64, 16, 70, 5, fp32=1) < 1e-4
    assert run(emul, oracle, 64, 8, 40, 4, fp32=1) < 1e-4


def test_geometry_of_the_shapes_the_engine_was_built_for(emul):
    def geom(F, T, Q, LT, SKW, nls, Pt):
        out = (C.c_long * 7)()
        emul.band_emul_geometry(F, T, Q, LT, SKW, nls, Pt, out)
        return dict(zip(("P", "gap", "LAG", "R", "nblk", "U", "rows"), out))
    # lws(2048,256): 1025 bins on 128 lanes 8 steps apart; a ring of 68 rows x 128 lanes x 8 B = 69.6 KB: two sweep slots
    g = geom(1025, 250, 8, 5, 8, 128, 8)
    assert (g["P"], g["gap"], g["LAG"], g["R"]) == (1032, 8, 72, 68) and 2 * g["R"] * 128 * 8 < 160 * 1024
    # lws(1024,64): sixteen frames per stencil row
    g = geom(513, 500, 16, 5, 8, 64, 16)
    assert (g["P"], g["gap"], g["LAG"], g["R"]) == (520, 8, 136, 132) and 2 * g["R"] * 64 * 8 < 160 * 1024
    # lws(1024,256, L=8) on the LT = 10 build
    g = geom(513, 500, 4, 10, 12, 64, 4)
    assert (g["P"], g["gap"], g["LAG"], g["R"]) == (768, 0, 48, 39)
    # a sweep ends with the last real frame: 500 + 14 frames on 128 lanes = four full blocks and two frames
    g = geom(1025, 500, 8, 5, 8, 128, 8)
    assert g["nblk"] == 5 and g["U"] == 8 * 1 + 1032 * 5 + 5 + 3
    for F, Q, LT, SKW, nls in ((1025, 8, 5, 8, 128), (513, 16, 5, 7, 64), (513, 4, 10, 12, 64), (33, 2, 5, 7, 64), (4097, 4, 5, 9, 512)):
        g = geom(F, 100, Q, LT, SKW, nls, Q)
        assert g["P"] % SKW == 0 and g["P"] >= F + LT and g["LAG"] % 2 == 0 and g["U"] % 2 == 0
        assert g["R"] >= 2 * LT + 1          # an image above Nyquist is read 2 LT rows back
