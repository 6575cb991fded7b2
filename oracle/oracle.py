"""ctypes loaders for the CPU checkers -- TEST INFRASTRUCTURE ONLY.

``Oracle``  : oracle/liblws_oracle.so, our fp64 C restatement (lws_oracle.c).
``RefLib``  : oracle/_ref/liblws_ref.so, the reference's own lwslib.cpp compiled in place by
              oracle/Makefile (C++-mangled symbols, argument order of lwslib.h:6-26).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblws_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "liblws_ref.so")
M0_ALL = 1 << 28
FLAVOUR_CANONICAL, FLAVOUR_NOFUTURE_Q4_COMPAT = 0, 1


def build(quiet=True):
    """(Re)build the checkers with oracle/Makefile (gcc only; skips _ref if the reference is absent)."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def split_weights(W):
    """(wr, wi, flag) exactly as lws.pyx:227-232 prepares them."""
    W = np.asarray(W)
    return (_f64(W.real), _f64(W.imag),
            np.ascontiguousarray(np.abs(W) > 1.0e-12, dtype=np.intc))


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = lib = C.CDLL(ORACLE_SO)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        lib.lwso_extend.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci]
        lib.lwso_extract.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci]
        lib.lwso_amplitude.argtypes = [vp, vp, vp, ci]
        lib.lwso_sweep.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, ci, cd]
        lib.lwso_online.argtypes = [vp] * 12 + [ci] * 7 + [cd, vp, ci]
        lib.lwso_repeat_kernel.argtypes = [vp, ci] + [vp] * 6 + [ci] * 4 + [cd, ci]
        lib.lwso_repeat_sweep.argtypes = [vp] * 6 + [ci] * 5 + [cd, ci]
        lib.lwso_batch_lws.argtypes = [vp, vp, ci, ci, vp, ci, ci, ci, vp, ci, cd]
        lib.lwso_nofuture_lws.argtypes = [vp, vp, ci, ci, vp, ci, ci, ci, vp, ci, cd, ci]
        lib.lwso_online_lws.argtypes = [vp, vp, ci, ci, vp, vp, vp, ci, ci, ci, vp, ci, ci, cd, cd]
        for f in (lib.lwso_batch_lws, lib.lwso_nofuture_lws, lib.lwso_online_lws):
            f.restype = ci

    # ---- kernel level -------------------------------------------------------------------
    def extend(self, S, L, Q):
        S = np.asarray(S, dtype=np.complex128)
        T, F = S.shape
        er = np.empty((T + 2 * (Q - 1), F + 2 * L))
        ei = np.empty_like(er)
        sr, si = _f64(S.real), _f64(S.imag)
        self.lib.lwso_extend(_ptr(er), _ptr(ei), _ptr(sr), _ptr(si), F, T, L, Q)
        return er, ei

    def sweep(self, er, ei, W, amp, F, M, L, Q, threshold, M0=M0_ALL, flavour=FLAVOUR_CANONICAL,
              update=2, qdiv=None, row0=0):
        """In-place sweep on extended buffers starting at extended row `row0` (pointer offset)."""
        wr, wi, wf = split_weights(W)
        Qp = W.shape[0]
        Np = F + 2 * L
        off = row0 * Np * 8
        amp = _f64(amp)
        self.lib.lwso_sweep(flavour, C.c_void_p(er.ctypes.data + off), C.c_void_p(ei.ctypes.data + off),
                            _ptr(wr), _ptr(wi), _ptr(wf), C.c_void_p(amp.ctypes.data + off), F, M, M0, L,
                            Q, Qp, float(threshold), update, float(Q if qdiv is None else qdiv))

    # ---- wrapper level (== lws.pyx:209-375) ------------------------------------------------
    @staticmethod
    def _prep(S, W):
        S = np.ascontiguousarray(S, dtype=np.complex128)
        W = np.ascontiguousarray(W, dtype=np.complex128)
        return S, W, np.empty_like(S)

    def batch_lws(self, S, W, thresholds, mean_amp=None):
        S, W, out = self._prep(S, W)
        thr = _f64(thresholds)
        T, F = S.shape
        Qp, Q, L1 = W.shape
        ma = float(np.mean(np.abs(S))) if mean_amp is None else mean_amp
        rc = self.lib.lwso_batch_lws(_ptr(S), _ptr(out), T, F, _ptr(W), L1 - 1, Q, Qp, _ptr(thr), thr.size, ma)
        if rc:
            raise ValueError('Please only include non-negative frequencies in the input spectrogram.')
        return out

    def nofuture_lws(self, S, W, thresholds, compat=True, mean_amp=None):
        S, W, out = self._prep(S, W)
        thr = _f64(thresholds)
        T, F = S.shape
        Qp, Q, L1 = W.shape
        ma = float(np.mean(np.abs(S))) if mean_amp is None else mean_amp
        rc = self.lib.lwso_nofuture_lws(_ptr(S), _ptr(out), T, F, _ptr(W), L1 - 1, Q, Qp, _ptr(thr),
                                        thr.size, ma, 1 if compat else 0)
        if rc:
            raise ValueError('Please only include non-negative frequencies in the input spectrogram.')
        return out

    def online_lws(self, S, W, W_ai, W_af, thresholds, LA, fshift, mean_amp=None):
        S, W, out = self._prep(S, W)
        W_ai = np.ascontiguousarray(W_ai, dtype=np.complex128)
        W_af = np.ascontiguousarray(W_af, dtype=np.complex128)
        thr = _f64(thresholds)
        T, F = S.shape
        Qp, Q, L1 = W.shape
        ma = float(np.mean(np.abs(S))) if mean_amp is None else mean_amp
        qdiv = float(2 * (F - 1) / fshift)
        rc = self.lib.lwso_online_lws(_ptr(S), _ptr(out), T, F, _ptr(W), _ptr(W_ai), _ptr(W_af), L1 - 1,
                                      Q, Qp, _ptr(thr), thr.size, int(LA), qdiv, ma)
        if rc:
            raise ValueError('Please only include non-negative frequencies in the input spectrogram.')
        return out


class RefLib:
    """The reference's kernels, called through their mangled C++ names."""
    SYMS = {
        "ExtendSpec": "_Z10ExtendSpecPdS_S_S_iiii",
        "CopySpec": "_Z8CopySpecPdS_S_S_iiii",
        "ComputeAmpSpec": "_Z14ComputeAmpSpecPdS_S_i",
        "LWSQ2": "_Z5LWSQ2PdS_S_S_PiS_iiid",
        "LWSQ4": "_Z5LWSQ4PdS_S_S_PiS_iiid",
        "LWSanyQ": "_Z7LWSanyQPdS_S_S_PiS_iiiid",
        "LWSfractionalQ": "_Z14LWSfractionalQPdS_S_S_PiS_iiiid",
        "NoFuture_LWSQ2": "_Z14NoFuture_LWSQ2PdS_S_S_PiS_iiid",
        "NoFuture_LWSQ4": "_Z14NoFuture_LWSQ4PdS_S_S_PiS_iiid",
        "NoFuture_LWSanyQ": "_Z16NoFuture_LWSanyQPdS_S_S_PiS_iiiid",
        "NoFuture_LWSfractionalQ": "_Z23NoFuture_LWSfractionalQPdS_S_S_PiS_iiiid",
        "Asym_UpdatePhaseQ2": "_Z18Asym_UpdatePhaseQ2PdS_S_S_PiS_iiiidi",
        "Asym_UpdatePhaseQ4": "_Z18Asym_UpdatePhaseQ4PdS_S_S_PiS_iiiidi",
        "Asym_UpdatePhaseanyQ": "_Z20Asym_UpdatePhaseanyQPdS_S_S_PiS_iiiiidi",
        "Asym_UpdatePhasefractionalQ": "_Z27Asym_UpdatePhasefractionalQPdS_S_S_PiS_iiiiiddi",
        "TF_RTISI_LA": "_Z11TF_RTISI_LAPdS_S_S_S_S_S_S_PiS0_S0_S_iiiiiidiS_i",
    }

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=None):
        # `path`: any library exporting the lwslib.h symbols (tests point it at liblws_hip.so's compat shims)
        self.lib = C.CDLL(path or REF_SO)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        six = [vp] * 6
        sig = {
            "ExtendSpec": [vp] * 4 + [ci] * 4, "CopySpec": [vp] * 4 + [ci] * 4,
            "ComputeAmpSpec": [vp] * 3 + [ci],
            "LWSQ2": six + [ci, ci, ci, cd], "LWSQ4": six + [ci, ci, ci, cd],
            "LWSanyQ": six + [ci, ci, ci, ci, cd], "LWSfractionalQ": six + [ci, ci, ci, ci, cd],
            "NoFuture_LWSQ2": six + [ci, ci, ci, cd], "NoFuture_LWSQ4": six + [ci, ci, ci, cd],
            "NoFuture_LWSanyQ": six + [ci, ci, ci, ci, cd],
            "NoFuture_LWSfractionalQ": six + [ci, ci, ci, ci, cd],
            "Asym_UpdatePhaseQ2": six + [ci, ci, ci, ci, cd, ci],
            "Asym_UpdatePhaseQ4": six + [ci, ci, ci, ci, cd, ci],
            "Asym_UpdatePhaseanyQ": six + [ci, ci, ci, ci, ci, cd, ci],
            "Asym_UpdatePhasefractionalQ": six + [ci, ci, ci, ci, ci, cd, cd, ci],
            "TF_RTISI_LA": [vp] * 12 + [ci] * 6 + [cd, ci, vp, ci],
        }
        self.fn = {}
        for name, sym in self.SYMS.items():
            f = getattr(self.lib, sym)
            f.argtypes = sig[name]
            f.restype = None
            self.fn[name] = f

    def extend(self, S, L, Q):
        S = np.asarray(S, dtype=np.complex128)
        T, F = S.shape
        er = np.empty((T + 2 * (Q - 1), F + 2 * L))
        ei = np.empty_like(er)
        sr, si = _f64(S.real), _f64(S.imag)
        self.fn["ExtendSpec"](_ptr(er), _ptr(ei), _ptr(sr), _ptr(si), F, T, L, Q)
        return er, ei

    def call(self, name, er, ei, W, amp, *tail, row0=0, Np=None):
        """name(Sr+row0*Np, Si+row0*Np, wr, wi, flag, amp+row0*Np, *tail)."""
        wr, wi, wf = split_weights(W)
        off = 0 if not row0 else row0 * Np * 8
        self.fn[name](C.c_void_p(er.ctypes.data + off), C.c_void_p(ei.ctypes.data + off), _ptr(wr), _ptr(wi),
                      _ptr(wf), C.c_void_p(amp.ctypes.data + off), *tail)
