"""Row a9: the lwslib.h-compatible shims exported by liblws_hip.so (include/lwslib_compat.h).

The shims carry the reference's own C++-mangled names, so the ctypes table written for the compiled reference
(oracle.RefLib) binds them unchanged.  Every family is checked against the golden vectors the reference produced
(tests/golden/sweeps.npz, wrappers.npz, general_weights.npz) and against the oracle on the same buffers.
fp64 on both sides: 1e-11 for one sweep, 1e-9 for drivers that chain sweeps (see tests/test_oracle.py).
"""
import ctypes as C

import numpy as np
import pytest

import lws_amd._capi as capi
from conftest import load_golden
from oracle.oracle import RefLib, split_weights, _ptr

pytestmark = pytest.mark.gpu
ATOL_SWEEP, ATOL_MULTI = 1e-11, 1e-9


@pytest.fixture(scope="module")
def shim():
    return RefLib(capi.LIB_PATH)


def _cases(g):
    for ci in range(int(g["ncases"])):
        tag = f"c{ci}"
        fsize, fshift, Q, T, F, L = [int(v) for v in g[f"{tag}_meta"]]
        yield tag, Q, T, F, L


def _fresh(shim, S, L, Q):
    er, ei = shim.extend(S, L, Q)
    amp = np.empty_like(er)
    shim.fn["ComputeAmpSpec"](_ptr(er), _ptr(ei), _ptr(amp), er.size)
    return er, ei, amp


def test_helpers_match_golden(shim, oracle):
    g = load_golden("helpers.npz")
    er, ei = shim.extend(g["ext_in"], 2, 3)
    assert np.array_equal(er + 1j * ei, g["ext_L2_Q3"])
    T, F = g["ext_in"].shape
    sr, si = np.zeros((T, F)), np.zeros((T, F))
    shim.fn["CopySpec"](_ptr(er), _ptr(ei), _ptr(sr), _ptr(si), F, T, 2, 3)
    assert np.array_equal(sr + 1j * si, g["ext_in"])
    amp = np.empty_like(er)
    shim.fn["ComputeAmpSpec"](_ptr(er), _ptr(ei), _ptr(amp), er.size)
    assert np.array_equal(amp, np.sqrt(er * er + ei * ei))


def test_single_sweeps_all_families(shim):
    g = load_golden("sweeps.npz")
    n = 0
    for tag, Q, T, F, L in _cases(g):
        Np = F + 2 * L
        S, W, W_ai, W_af = g[f"{tag}_S"], g[f"{tag}_W"], g[f"{tag}_W_ai"], g[f"{tag}_W_af"]
        sets = {"W": W, "W_ai": W_ai, "W_af": W_af}
        qname = {2: "Q2", 4: "Q4"}.get(Q)
        for ti, thr in enumerate(g[f"{tag}_thr"]):
            thr = float(thr)
            er, ei, amp = _fresh(shim, S, L, Q)
            shim.call("LWSanyQ", er, ei, W, amp, F, T, L, Q, thr)
            assert np.abs(er + 1j * ei - g[f"{tag}_t{ti}_batch_any"]).max() < ATOL_SWEEP
            if qname:
                er, ei, amp = _fresh(shim, S, L, Q)
                shim.call("LWS" + qname, er, ei, W, amp, F, T, L, thr)
                assert np.abs(er + 1j * ei - g[f"{tag}_t{ti}_batch_q"]).max() < ATOL_MULTI
            for wname in ("W", "W_ai"):
                er, ei, amp = _fresh(shim, S, L, Q)
                shim.call("NoFuture_LWSanyQ", er, ei, sets[wname], amp, F, T, L, Q, thr)
                assert np.abs(er + 1j * ei - g[f"{tag}_t{ti}_nofut_any_{wname}"]).max() < ATOL_SWEEP
                key = f"{tag}_t{ti}_nofut_q_{wname}"
                if qname and key in g:  # Q4: the shipped flat-offset addressing, reproduced
                    er, ei, amp = _fresh(shim, S, L, Q)
                    shim.call("NoFuture_LWS" + qname, er, ei, sets[wname], amp, F, T, L, thr)
                    assert np.abs(er + 1j * ei - g[key]).max() < ATOL_MULTI, key
            for ai, (row0, M, M0, wi) in enumerate(g[f"{tag}_asym_shapes"]):
                for upd in (2, 1):
                    key = f"{tag}_t{ti}_asym{ai}_u{upd}"
                    if key not in g:
                        continue
                    Wsel = [W, W_ai, W_af][wi]
                    er, ei, amp = _fresh(shim, S, L, Q)
                    shim.call("Asym_UpdatePhaseanyQ", er, ei, Wsel, amp, F, int(M), int(M0), L, Q, thr, upd,
                              row0=int(row0), Np=Np)
                    assert np.abs(er + 1j * ei - g[key]).max() < ATOL_SWEEP, key
                    if qname:
                        er, ei, amp = _fresh(shim, S, L, Q)
                        shim.call("Asym_UpdatePhase" + qname, er, ei, Wsel, amp, F, int(M), int(M0), L, thr, upd,
                                  row0=int(row0), Np=Np)
                        assert np.abs(er + 1j * ei - g[key]).max() < ATOL_MULTI, key
                    n += 1
    assert n > 20


def test_infinite_threshold_is_identity(shim):
    g = load_golden("sweeps.npz")
    for tag, Q, T, F, L in _cases(g):
        er, ei, amp = _fresh(shim, g[f"{tag}_S"], L, Q)
        e0 = er + 1j * ei
        shim.call("LWSanyQ", er, ei, g[f"{tag}_W"], amp, F, T, L, Q, 1e30)
        assert np.array_equal(er + 1j * ei, e0)


def _online(shim, S, W, W_ai, W_af, thr, LA, qfloat, summarized, update=2):
    Qp, Q, L1 = W.shape
    L = L1 - 1
    T, F = S.shape
    er, ei, amp = _fresh(shim, S, L, Q)
    mean = np.abs(S).mean()
    th = np.ascontiguousarray(np.asarray(thr, dtype=np.float64) * mean)
    w = [split_weights(x) for x in (W, W_ai, W_af)]
    shim.fn["TF_RTISI_LA"](_ptr(er), _ptr(ei), _ptr(w[0][0]), _ptr(w[0][1]), _ptr(w[1][0]), _ptr(w[1][1]),
                           _ptr(w[2][0]), _ptr(w[2][1]), _ptr(w[0][2]), _ptr(w[1][2]), _ptr(w[2][2]), _ptr(amp),
                           th.size, LA, F, T, L, Q, float(qfloat), int(summarized), _ptr(th), update)
    sr, si = np.zeros((T, F)), np.zeros((T, F))
    shim.fn["CopySpec"](_ptr(er), _ptr(ei), _ptr(sr), _ptr(si), F, T, L, Q)
    return sr + 1j * si


@pytest.mark.parametrize("tag", ["64_16", "64_32", "64_8", "48_16"])
def test_online_driver(shim, tag):
    g, h = load_golden("wrappers.npz"), load_golden("helpers.npz")
    S, thr = g[f"S_{tag}"], g[f"thr_{tag}"]
    W, W_ai, W_af = h[f"W_{tag}"], h[f"W_ai_{tag}"], h[f"W_af_{tag}"]
    fsize, fshift = [int(v) for v in tag.split("_")]
    q = fsize / fshift
    assert np.abs(_online(shim, S, W, W_ai, W_af, thr[:3], 3, q, 1) - g[f"online_{tag}"]).max() < ATOL_MULTI
    assert np.abs(_online(shim, S, W, W_ai, W_af, thr[:3], 0, q, 1) - g[f"online_la0_{tag}"]).max() < ATOL_MULTI
    assert np.abs(_online(shim, S, W, W_ai, W_af, thr[:2], 5, q, 1) - g[f"online_la5_{tag}"]).max() < ATOL_MULTI


@pytest.mark.parametrize("tag", ["32_8", "32_12"])
def test_fractional_kernels_periodic_row(shim, tag):
    """fractionalQ family with general [N][Q][L+1] weights; the reference was pinned with a periodic extra row."""
    g = load_golden("general_weights.npz")
    fsize, fshift, T, F, Q, L, LA = [int(v) for v in g[f"meta_{tag}"]]
    S, thr = g[f"S_{tag}"], float(g[f"thr_{tag}"][0])
    W, W_ai, W_af = g[f"W_{tag}"], g[f"W_ai_{tag}"], g[f"W_af_{tag}"]
    er, ei, amp = _fresh(shim, S, L, Q)
    shim.call("LWSfractionalQ", er, ei, W, amp, F, T, L, Q, thr)
    shim.call("LWSfractionalQ", er, ei, W, amp, F, T, L, Q, 0.0)
    assert np.abs(er + 1j * ei - g[f"batch_{tag}"]).max() < ATOL_MULTI
    er, ei, amp = _fresh(shim, S, L, Q)
    shim.call("NoFuture_LWSfractionalQ", er, ei, W_ai, amp, F, T, L, Q, thr)
    assert np.abs(er + 1j * ei - g[f"nofuture_{tag}"]).max() < ATOL_MULTI


def test_asym_fractional_matches_oracle(shim, oracle):
    g = load_golden("general_weights.npz")
    tag = "32_12"
    fsize, fshift, T, F, Q, L, LA = [int(v) for v in g[f"meta_{tag}"]]
    S, W = g[f"S_{tag}"], g[f"W_{tag}"]
    qf = fsize / fshift
    for M, M0, row0, upd in [(3, 4, 1, 2), (1, 0, 2, 2), (4, 2, 0, 1)]:
        er, ei, amp = _fresh(shim, S, L, Q)
        e2, i2 = er.copy(), ei.copy()
        shim.call("Asym_UpdatePhasefractionalQ", er, ei, W, amp, F, M, M0, L, Q, qf, 0.1, upd, row0=row0,
                  Np=F + 2 * L)
        oracle.sweep(e2, i2, W, amp, F, M, L, Q, 0.1, M0=M0, update=upd, qdiv=qf, row0=row0)
        assert np.abs(er + 1j * ei - (e2 + 1j * i2)).max() < ATOL_SWEEP
