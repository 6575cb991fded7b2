"""The schedule of the fp64 systolic engine (lws_amd/csrc/lws_sys64.hip) on the CPU: tools/sys64_model.py steps through it lane
for lane in numpy -- same lane/frame mapping, ring rows, scatter order, image handling, frame period and lag as the kernel -- and
must reproduce the oracle.  Runs without a GPU; the kernel itself is tested in tests/test_gpu_sys64.py."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import sys64_model as model  # noqa: E402

import lws_amd  # noqa: E402


@pytest.fixture(scope="module")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.mark.parametrize("fsize,fshift,T,iters", [(64, 16, 9, 4),      # Q = 4, one block of frames, two passes of three slots
                                                   (64, 32, 67, 4),     # Q = 2, lanes wrap into a second block
                                                   (1012, 253, 5, 2)])  # 507 bins: the last images are written in the next frame's steps
def test_model_reproduces_the_oracle(fsize, fshift, T, iters, oracle):
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    W = lws_amd.lws(fsize, fshift).W
    S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
    thr = lws_amd.get_thresholds(iters, 1.0, 0.1, 1)
    ref = oracle.batch_lws(S, W, thr)
    out, ages = model.batch_lws_model(S, W, thr, NS=3)
    assert np.abs(out - ref).max() < 1e-12 * np.abs(ref).max()
    P, gap, LAG, R = model.geometry(F, W.shape[1])
    # what the kernel's rings rely on: a slot reads the slot before it at least 2 steps and at most R - 1 steps back,
    # and itself at most R - 1 steps back
    assert ages["prev_min"] >= 2 and ages["prev_max"] <= R - 1 and ages["own_max"] <= R - 1 and ages["own_min"] >= 2
    assert LAG % 8 == 0 and P % 8 == 0 and P >= F + model.L


def test_geometry_matches_the_kernel_for_the_headline_shape():
    # lws_sys64.hip: geom(): 513 bins -> period 520, lag 40, ring 36 rows -> four sweep slots in 160 KB
    assert model.geometry(513, 4) == (520, 8, 40, 36)
    assert model.geometry(257, 4) == (512, 0, 32, 28)
    assert model.geometry(513, 2) == (520, 8, 24, 20)


def test_prefetch_stays_inside_the_scratch_rows():
    """lws_sys64.hip's slot 0 prefetches the neighbour frames of the next steps unconditionally; in lanes whose right-hand neighbour
    wrapped into the next block of frames the row is `gap` further on.  The rows after the skewed state are sized from the geometry
    (round 4 had a fixed margin of 96 rows: frames of more than ~570 bins read up to 57 KB past a spectrogram's scratch -- past the
    allocation for the last one).  Host-side check over every frame length the engine takes (no GPU needed)."""
    import ctypes as C
    from lws_amd import _capi
    lib = _capi.load_raw()
    lib.lws_debug_sys64_layout.argtypes = [C.c_int] * 3 + [C.c_void_p]
    seen, worst = 0, None
    for Q in (2, 4):
        for F in range(17, 2200, 2):
            for T in (1, 63, 500, 3000):
                out = (C.c_long * 4)()
                if not lib.lws_debug_sys64_layout(F, T, Q, out):
                    continue
                rows, hi_read, hi_write, gap = out
                seen += 1
                assert hi_read < rows and hi_write < rows, (Q, F, T, list(out))
                if worst is None or rows - 1 - hi_read < worst[0]:
                    worst = (rows - 1 - hi_read, Q, F, T, gap)
    assert seen > 3000 and worst[0] >= 0
    out = (C.c_long * 4)()
    # (round 4's case -- lws(1200,300), 601 bins, a gap of 96 steps on the 64-lane geometry -- now runs with 128 frames in flight and no
    #  gap at all; the longest frames of that geometry are the ones with surplus steps)
    assert lib.lws_debug_sys64_layout(601, 90, 4, out) and out[3] == 0
    assert lib.lws_debug_sys64_layout(1051, 90, 4, out) and out[3] == 32
    assert lib.lws_debug_sys64_layout(1101, 90, 4, out) and out[3] == 0      # 256 frames in flight
    assert lib.lws_debug_sys64_layout(2049, 90, 4, out) and out[3] == 8
    assert not lib.lws_debug_sys64_layout(2101, 90, 4, out)
