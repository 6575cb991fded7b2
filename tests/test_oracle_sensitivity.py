"""CPU: how far can ANY fp32 engine be from the reference after the online stage?  The reference's own arithmetic (the fp64 oracle,
pinned to the reference in tests/test_oracle.py) is run twice on BASELINE config 3's input: on the no-future result as is, and on
the same values rounded to complex64 -- a perturbation of one fp32 ulp, the least an engine that stores fp32 state can commit.
TF_RTISI_LA (lwslib.cpp:1424-1492) re-projects every frame from its predecessors 41 times; the difference grows about tenfold
every 16-20 frames and is O(1) after ~120 frames.  tests/test_gpu_parity.py::test_config3_tolerance_stage_by_stage asserts the
fp32 engines against this envelope (value-level on the first frames, no further than this from the oracle afterwards)."""
import numpy as np

import lws_amd


def test_online_stage_amplifies_a_one_ulp_fp32_perturbation(oracle):
    rng = np.random.default_rng(20260928 + 3)
    T = 160
    M = np.abs(rng.standard_normal((T, 513)) + 1j * rng.standard_normal((T, 513))).astype(np.float32).astype(np.float64)
    awin = np.sqrt(lws_amd.hann(1024, symmetric=True, use_offset=False))
    awin = np.sqrt(awin * lws_amd.synthwin(awin, 256))
    swin = lws_amd.synthwin(awin, 256)
    W = lws_amd.create_weights(awin, swin, 256, 5)
    win_ai, win_af = lws_amd.build_asymmetric_windows(awin * swin, 256)
    W_ai, W_af = lws_amd.create_weights(win_ai, swin, 256, 5), lws_amd.create_weights(win_af, swin, 256, 5)
    thr = lws_amd.get_thresholds(10, 1, 0.1, 1)
    for compat in (True, False):
        r0 = oracle.nofuture_lws(M, W_ai, [1.0], compat=compat)
        r1 = oracle.online_lws(r0, W, W_ai, W_af, thr, 3, 256)
        r1p = oracle.online_lws(r0.astype(np.complex64).astype(np.complex128), W, W_ai, W_af, thr, 3, 256)
        err = np.linalg.norm(r1p - r1, axis=1) / np.linalg.norm(r1, axis=1)      # per frame
        assert err[:8].max() < 1e-5, err[:8]                   # the perturbation itself: ~1e-7
        assert err[:32].max() < 1e-3, err[:32].max()
        assert np.median(err[128:]) > 0.1, np.median(err[128:])     # decorrelated: O(1)
        assert err[64:96].mean() > 10 * err[16:32].mean()      # and growing in between
