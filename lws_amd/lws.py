"""Python surface of the reference's ``lws`` module (python/lws.pyx, v1.2.8) on the MI355X engine.

Same names, arguments, defaults, return dtypes and error behaviour as the reference; the three LWS
entry points ``batch_lws`` / ``nofuture_lws`` / ``online_lws`` (lws.pyx:209-375) and the methods of
``class lws`` call the HIP engine through the C ABI (include/lws_hip.h) instead of the CPU kernels of
lwslib.cpp.  Everything else in this file is host-side setup (windows, weights, STFT for evaluation):
one-off numpy work that feeds the kernels, restated here so the package is self-contained.

Extensions that do not change 2-D behaviour: the LWS entry points also accept a ``(B, T, F)`` stack of
independent spectrograms, and ``class lws`` takes ``device=`` / ``precision=`` / ``nofuture_q4_compat=``.
"""
from __future__ import annotations

import collections
import hashlib
import threading
import math

import numpy as np

from . import _capi

__version__ = "1.2.8"  # version of the interface mirrored


# --------------------------------------------------------------------------------------------
# windows (lws.pyx:10-40)
# --------------------------------------------------------------------------------------------
def hann(n, symmetric=True, use_offset=False):
    """Hann window of length n (lws.pyx:10-19)."""
    if symmetric:
        phase = (2.0 * np.arange(n) + 1.0) / (2.0 * n)  # sample points at the half-integers
    else:
        phase = (np.arange(n) + (1 if use_offset else 0)) / float(n)
    return 0.5 * (1.0 - np.cos(2.0 * np.pi * phase))


def synthwin(awin, fshift, swin=None):
    """Synthesis window normalised for perfect reconstruction (lws.pyx:22-40)."""
    fsize = len(awin)
    Q = int(math.ceil(float(fsize) / float(fshift)))
    if swin is None:
        swin = awin
    prod = np.zeros(Q * fshift)
    prod[:fsize] = np.asarray(awin) * np.asarray(swin)
    overlap = prod.reshape(Q, fshift).sum(axis=0)  # sum of the shifted window products
    norm = np.tile(overlap, Q)[:fsize]
    if norm.min() <= 0:
        raise ValueError('The normalizer is not strictly positive')
    return swin / norm


# --------------------------------------------------------------------------------------------
# STFT / iSTFT / consistency (lws.pyx:43-144) -- evaluation helpers, host numpy
# --------------------------------------------------------------------------------------------
def _perfectrec_prepad(fsize, fshift):
    rem = fsize % fshift
    return fsize - fshift if rem == 0 else fsize - rem


def stft(x, fsize, fshift, awin, fftsize=None, perfectrec=False):
    """STFT with a fixed frame shift; returns (frames, fftsize//2+1) complex128 (lws.pyx:43-90)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError('We only deal with single channel signals here')
    if fftsize is None:
        fftsize = fsize
    if fftsize % 2 == 1:
        raise ValueError('Odd ffts not supported.')
    x = x.astype(np.float64, copy=False) if not np.iscomplexobj(x) else x
    if perfectrec is True:
        pre = _perfectrec_prepad(fsize, fshift)
        post = (-len(x)) % fshift
        x = np.concatenate([np.zeros(pre), x, np.zeros(post)])
        M = len(x) // fshift
    else:
        post = (-(len(x) - fsize)) % fshift
        x = np.concatenate([x, np.zeros(post)])
        M = (len(x) - fsize) // fshift + 1
    need = (M - 1) * fshift + fsize
    if need > len(x):
        x = np.concatenate([x, np.zeros(need - len(x))])
    idx = fshift * np.arange(M)[:, None] + np.arange(fsize)[None, :]
    frames = x[idx] * np.asarray(awin)[None, :]
    spec = np.fft.fft(frames, n=fftsize, axis=1)[:, : fftsize // 2 + 1]
    return spec.astype(np.complex128)


def istft(spec, fshift, swin, awin=None, fftsize=None, perfectrec=False):
    """Inverse STFT by weighted overlap-add (lws.pyx:93-137)."""
    spec = np.asarray(spec)
    if spec.ndim != 2:
        raise ValueError('We only deal with single channel signals here')
    M, N = spec.shape
    if N % 2 != 1:
        raise ValueError('We expect the spectrogram to only have non-negative frequencies')
    fsize = 2 * (N - 1)
    if awin is not None:
        swin = synthwin(awin, fshift, swin=swin)
    if fftsize is None:
        fftsize = fsize
    swin = np.asarray(swin, dtype=np.float64)
    if fftsize > len(swin):
        swin = np.concatenate([swin, np.zeros(fftsize - len(swin))])
    full = np.concatenate([spec, np.conjugate(spec[:, -2:0:-1])], axis=1)  # Hermitian completion
    # (as in the reference, lws.pyx:107-126: the frame is 2 (bins - 1) samples, so a window of any other length -- which is what
    # every fftsize > fsize makes of it, and an fftsize < fsize of the frame -- fails to broadcast: ValueError)
    frames = np.real(np.fft.ifft(full, n=fftsize, axis=1))[:, :fsize] * np.squeeze(swin)[None, :]
    if frames.shape[1] != fsize:
        raise ValueError('operands could not be broadcast together with shapes (%d,) (%d,)' % (fsize, frames.shape[1]))
    signal = np.zeros(fshift * (M - 1) + fsize)
    for s in range(M):
        signal[fshift * s: fshift * s + fsize] += frames[s]
    if perfectrec is True:
        signal = signal[_perfectrec_prepad(fsize, fshift):(fshift - fsize)]
    return signal


def get_consistency(S, fsize, fshift, awin, swin, perfectrec=False):
    """20 log10(|S| / |stft(istft(S)) - S|) in dB (lws.pyx:140-144)."""
    back = stft(istft(S, fshift, swin, perfectrec=perfectrec), fsize, fshift, awin, perfectrec=perfectrec)
    return 20 * np.log10(np.linalg.norm(S) / np.linalg.norm(back - S))


def _dev_tensor(a, dtype, device):
    import torch
    dev = torch.device("cuda", int(device))
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=dev, dtype=dtype).contiguous()


def stft_dev(x, fsize, fshift, awin, fftsize=None, perfectrec=False, device=0):
    """stft() above on the device (lws_stft.hip, float32): x a signal (len,) or a stack (B, len), numpy or torch; returns a complex64
    torch tensor (T, fftsize//2+1) / (B, T, ...).  fftsize > fsize: the fsize windowed samples followed by zeros, as
    np.fft.fft(frame, n=fftsize) does (lws.pyx:49-50,85); any even fftsize in [32, 4096]."""
    import torch
    if fftsize is None:
        fftsize = fsize
    if fftsize % 2 == 1:
        raise ValueError('Odd ffts not supported.')
    t = _dev_tensor(x, torch.float32, device)
    if t.dim() not in (1, 2):
        raise ValueError('We only deal with single channel signals here')
    single = t.dim() == 1
    if single:
        t = t[None]
    B, n = t.shape
    T = _capi.stft_frames(n, fsize, fshift, perfectrec)
    out = torch.empty((B, T, fftsize // 2 + 1), dtype=torch.complex64, device=t.device)
    _capi.stft_dev(t.data_ptr(), B, n, fsize, fshift, np.asarray(awin, dtype=np.float64), perfectrec, out.data_ptr(), device=int(device),
                   stream=torch.cuda.current_stream(t.device).cuda_stream, fftsize=fftsize)
    return out[0] if single else out


def istft_dev(spec, fshift, swin, awin=None, fftsize=None, perfectrec=False, device=0):
    """istft() above on the device: spec (T, F) or (B, T, F), numpy or torch; returns a float32 torch tensor (len,) / (B, len).
    As in the reference (lws.pyx:107-126) the frame is 2 (F - 1) samples and any other fftsize / window length is a ValueError."""
    import torch
    t = _dev_tensor(spec, torch.complex64, device)
    if t.dim() not in (2, 3):
        raise ValueError('We only deal with single channel signals here')
    single = t.dim() == 2
    if single:
        t = t[None]
    B, T, N = t.shape
    if N % 2 != 1:
        raise ValueError('We expect the spectrogram to only have non-negative frequencies')
    fsize = 2 * (N - 1)
    if awin is not None:
        swin = synthwin(awin, fshift, swin=swin)
    swin = np.squeeze(np.asarray(swin, dtype=np.float64))
    if (fftsize is not None and fftsize != fsize) or swin.shape != (fsize,):
        raise ValueError('operands could not be broadcast together with shapes (%d,) (%d,)' % (fsize, max(swin.size, fftsize or 0)))
    n = _capi.istft_length(T, fsize, fshift, perfectrec)
    out = torch.empty((B, n), dtype=torch.float32, device=t.device)
    _capi.istft_dev(t.data_ptr(), B, T, fsize, fshift, swin, perfectrec, out.data_ptr(), device=int(device),
                    stream=torch.cuda.current_stream(t.device).cuda_stream)
    return out[0] if single else out


def extspec(S, L, Q):
    """Extended spectrogram: L Hermitian columns each side, Q-1 repeated frames each end (lws.pyx:146-157)."""
    S = np.asarray(S)
    T, F = S.shape
    E = np.zeros((T + 2 * (Q - 1), F + 2 * L), dtype=S.dtype)
    E[Q - 1:Q - 1 + T, L:L + F] = S
    for j in range(1, L + 1):
        E[:, L - j] = np.conjugate(E[:, L + j])
        E[:, F + L - 1 + j] = np.conjugate(E[:, F + L - 1 - j])
    E[:Q - 1] = E[Q - 1]
    E[Q - 1 + T:] = E[Q - 2 + T]
    return E


# --------------------------------------------------------------------------------------------
# weights (lws.pyx:160-206)
# --------------------------------------------------------------------------------------------
def create_weights(awin, swin, fshift, L, use_summarized_weights=True):
    """Complex LWS weights, shape (Qprime, Q, L+1) (lws.pyx:160-181).

    W[p, q, l] = exp(2j*pi*p*q/Qf) * exp(-2j*pi*l*q/Qf) * sum_t awin[t]*swin[t+q*fshift]/T * exp(-2j*pi*l*t/T)
    with 1 subtracted from the (q=0, l=0) term; Qf = T/fshift; Qprime = Q when the shift divides the
    window and summarised weights are requested, else T.
    """
    awin = np.asarray(awin, dtype=np.float64)
    swin = np.asarray(swin, dtype=np.float64)
    T = len(awin)
    Q = int(math.ceil(float(T) / float(fshift)))
    Qf = float(T) / float(fshift)
    Qprime = Q if (T % fshift == 0 and use_summarized_weights) else T
    lag = np.arange(L + 1)[:, None]
    prod = np.zeros((T, Q))
    for q in range(Q):
        n = T - q * fshift
        prod[:n, q] = awin[:n] * swin[q * fshift:q * fshift + n] / T
    base = np.exp(-2j * np.pi * lag * np.arange(T)[None, :] / T).dot(prod)      # (L+1, Q)
    base = base * np.exp(-2j * np.pi * lag * np.arange(Q)[None, :] / Qf)
    base[0, 0] -= 1
    twiddle = np.exp(2j * np.pi * np.arange(Qprime)[:, None] * np.arange(Q)[None, :] / Qf)  # (Qprime, Q)
    W = base[:, None, :] * twiddle[None, :, :]                                  # (L+1, Qprime, Q)
    return np.ascontiguousarray(W.transpose(1, 2, 0))


def build_asymmetric_windows(awin_swin, fshift):
    """Mirrored envelopes of RTISI-LA from the analysis*synthesis product (lws.pyx:184-200)."""
    awin_swin = np.asarray(awin_swin, dtype=np.float64)
    T = len(awin_swin)
    Q = int(math.ceil(float(T) / float(fshift)))
    shifted = np.zeros((T, Q))
    for q in range(Q):
        n = T - q * fshift
        shifted[:n, q] = awin_swin[q * fshift:q * fshift + n]
    win_ai = shifted[:, 1:].sum(axis=1)[::-1]
    win_af = shifted.sum(axis=1)[::-1]
    if T % fshift == 2:  # kept verbatim from lws.pyx:198 (the MATLAB binding tests Q == 2 instead)
        win_ai = awin_swin
    return win_ai, win_af


def get_thresholds(iterations, alpha, beta, gamma):
    """alpha * exp(-beta * i**gamma), i = 0..iterations-1 (lws.pyx:203-206)."""
    return alpha * np.exp(-beta * np.arange(iterations) ** gamma)


# --------------------------------------------------------------------------------------------
# the hot path (lws.pyx:209-375) -> HIP engine
# --------------------------------------------------------------------------------------------
_PLAN_DEFAULTS = {"device": 0, "precision": "fp32", "nofuture_q4_compat": True, "force_generic": False, "storage": "fp32"}

# The reference's module-level functions take the weights with every call (lws.pyx:209,261,314) and user code calls them
# in loops over spectrograms; a device plan (weight upload, table analysis, scratch, events) per call would dominate small
# calls, so the last few plans are kept, keyed by the weights' bytes and the plan options.  `clear_plan_cache()` releases
# their device memory.  Precision: the engine computes in fp32 by default (`precision="fp64"` selects the reference's
# arithmetic: batch sweeps of Q = 2 / 4 plans on the fp64 systolic engine, lws_sys64.hip, everything else -- and everything
# with `force_generic=True` -- on the order-exact generic engine; tolerances in DESIGN.md section 6); complex128 in, complex128
# out either way.
_PLAN_CACHE_SIZE = 4
# One cache per calling thread: the bindings release the GIL around the C call and an lws_plan (scratch, events, stage
# buffers) serves one call at a time, so a plan shared between threads could be used -- or evicted and destroyed -- while
# another thread is inside the library with it.  A thread's plans die with the thread (Plan.__del__).
_plan_tls = threading.local()


def _thread_cache():
    cache = getattr(_plan_tls, "cache", None)
    if cache is None:
        cache = _plan_tls.cache = collections.OrderedDict()
    return cache


def _cached_plan(F, Ws, plan_kw):
    kw = {**_PLAN_DEFAULTS, **plan_kw}
    key = (int(F),) + tuple(None if w is None else (w.shape, hashlib.sha1(np.ascontiguousarray(w, dtype=np.complex128).tobytes()).hexdigest())
                            for w in Ws) + tuple(sorted(kw.items()))
    cache = _thread_cache()
    plan = cache.pop(key, None)
    if plan is None:
        plan = _capi.Plan(F, *Ws, **kw)
        while len(cache) >= _PLAN_CACHE_SIZE:
            cache.popitem(last=False)[1].close()   # (this thread's own plan, and it is not inside a call)
    cache[key] = plan          # most recently used last
    return plan


def clear_plan_cache():
    """Destroy the calling thread's plans kept for the module-level batch_lws / nofuture_lws / online_lws (frees their
    device memory)."""
    cache = _thread_cache()
    while cache:
        cache.popitem()[1].close()


def _prepare(S, W, use_simplifications, n_extra_w=()):
    """Argument handling shared by the three wrappers (lws.pyx:212-224)."""
    S = np.asarray(S)
    if S.dtype != np.complex128:
        S = S.astype(np.complex128)
    W = np.asarray(W)
    Qp, Q = W.shape[0], W.shape[1]
    F = S.shape[-1]
    return S, W, Qp, Q, F


def _check(S, W, Qp, Q, F, use_simplifications):
    if F % 2 == 0:
        raise ValueError('Please only include non-negative frequencies in the input spectrogram.')
    if Qp != Q and Qp != 2 * (F - 1):
        raise ValueError('Weights have %d rows: expected Q=%d (summarized) or N=%d (general).' % (Qp, Q, 2 * (F - 1)))
    if Qp == Q and not use_simplifications:
        # the reference would index a Q-row tensor with the bin number here (undefined behaviour)
        raise ValueError('use_simplifications=False needs general weights '
                         '(create_weights(..., use_summarized_weights=False)).')


def batch_lws(S, W, thresholds, use_simplifications=True, **plan_kw):
    """Batch LWS (lws.pyx:209-258): ``len(thresholds)`` in-place Gauss-Seidel sweeps."""
    S, W, Qp, Q, F = _prepare(S, W, use_simplifications)
    if len(thresholds) == 0:
        return S
    _check(S, W, Qp, Q, F, use_simplifications)
    return _cached_plan(F, (W, None, None), plan_kw).batch(S, thresholds)


def nofuture_lws(S, W, thresholds, use_simplifications=True, **plan_kw):
    """LWS using past frames only (lws.pyx:261-311); for Q == 4 the default reproduces the
    reference's NoFuture_LWSQ4 addressing (pass ``nofuture_q4_compat=False`` for the anyQ semantics)."""
    S, W, Qp, Q, F = _prepare(S, W, use_simplifications)
    if len(thresholds) == 0:
        return S
    _check(S, W, Qp, Q, F, use_simplifications)
    return _cached_plan(F, (W, None, None), plan_kw).nofuture(S, thresholds)


def online_lws(S, W, W_ai, W_af, thresholds, LA, fshift, use_simplifications=True, **plan_kw):
    """Online LWS / TF-RTISI-LA (lws.pyx:314-375)."""
    thresholds = np.asarray(thresholds, dtype=np.float64)
    if thresholds.ndim != 1:
        raise ValueError('Buffer has wrong number of dimensions (expected 1, got %d)' % thresholds.ndim)
    S, W, Qp, Q, F = _prepare(S, W, use_simplifications)
    if len(thresholds) == 0:
        return S
    _check(S, W, Qp, Q, F, use_simplifications)
    qdiv = float(2 * (F - 1) / int(fshift))  # lws.pyx:339
    return _cached_plan(F, (W, np.asarray(W_ai), np.asarray(W_af)), plan_kw).online(S, thresholds, int(LA), qdiv)


# ---- construction of an lws object: helpers ------------------------------------------------------------------------------
# (the keyword surface, defaults, printed notices and attribute names are the reference's interface, lws.pyx:379-455)
_STAGES = ("nofuture", "online", "batch")
_SCHEDULE_KEYS = ("iterations", "alpha", "beta", "gamma")
# what `mode` overrides (lws.pyx:437-442): stage -> iterations
_MODES = {None: {}, "speech": {"nofuture": 0, "online": 0}, "music": {"nofuture": 1, "online": 10}}


def _resolve_windows(awin_or_fsize, fshift, swin, fftsize, symmetric_win):
    """Analysis window (a frame size stands for the reference's default window: the square root of a Hann window,
    renormalised with its synthesis partner) zero-padded symmetrically to `fftsize`, and the synthesis window the caller
    passed, padded the same way.  Returns (awin, swin or None, padded samples per side)."""
    if isinstance(awin_or_fsize, (int, np.integer)):
        root = np.sqrt(hann(int(awin_or_fsize), symmetric=symmetric_win, use_offset=False))
        awin = np.sqrt(root * synthwin(root, fshift))
    else:
        awin = np.asarray(awin_or_fsize)
        if awin.ndim > 1:   # (lws.pyx:391 compares a shape tuple with an int -- a TypeError on Python 3; the intent is clear)
            if awin.ndim > 2 or min(awin.shape) > 1:
                raise ValueError('The analysis window should be flat')
            awin = awin.ravel()
    extra = 0 if fftsize is None else int(fftsize) - len(awin)
    if extra <= 0:
        return awin, swin, 0
    if extra % 2:
        raise ValueError('The zero-padding should add even length to the original window.')
    side = np.zeros(extra // 2)
    return np.hstack((side, awin, side)), (None if swin is None else np.hstack((side, swin, side))), extra // 2


class lws(object):
    """Configuration object of the reference (lws.pyx:378-499), same keyword arguments."""

    def __init__(self, awin_or_fsize, fshift, L=5, swin=None, look_ahead=3,
                 nofuture_iterations=0, nofuture_alpha=1, nofuture_beta=0.1, nofuture_gamma=1,
                 online_iterations=0, online_alpha=1, online_beta=0.1, online_gamma=1,
                 batch_iterations=100, batch_alpha=100, batch_beta=0.1, batch_gamma=1,
                 symmetric_win=True, mode=None, fftsize=None, perfectrec=True, use_simplifications=True,
                 device=0, precision="fp32", nofuture_q4_compat=True, force_generic=False, storage="fp32"):
        schedule = {"nofuture": (nofuture_iterations, nofuture_alpha, nofuture_beta, nofuture_gamma),
                    "online": (online_iterations, online_alpha, online_beta, online_gamma),
                    "batch": (batch_iterations, batch_alpha, batch_beta, batch_gamma)}
        # windows
        awin, swin_given, padded = _resolve_windows(awin_or_fsize, fshift, swin, fftsize, symmetric_win)
        if padded:
            print('Zero-padding symmetrically around the original windows.\n'
                  'WARNING: for code simplicity, a consequence is that the first/last '
                  '{} samples of the signal will not be '.format(padded) +
                  'in the perfect reconstruction region.')
        if swin_given is not None:
            print('Provided synthesis window is renormalized for perfect reconstruction.')
        self.awin, self.fshift, self.fsize = awin, fshift, len(awin)
        self.swin = synthwin(awin, fshift, swin=swin_given)
        self.perfectrec, self.L, self.use_simplifications, self.look_ahead = perfectrec, L, use_simplifications, look_ahead
        q = self.fsize / self.fshift
        self.Q = int(q) if self.fsize % self.fshift == 0 else q
        # weights of the three sweeps: symmetric windows, and the two asymmetric envelopes of RTISI-LA
        self.win_ai, self.win_af = build_asymmetric_windows(self.awin * self.swin, self.fshift)
        for attr, win in (("W", self.awin), ("W_ai", self.win_ai), ("W_af", self.win_af)):
            setattr(self, attr, create_weights(win, self.swin, self.fshift, self.L, use_summarized_weights=use_simplifications))
        # schedules: <stage>_iterations / _alpha / _beta / _gamma; `mode` presets the first two stages
        if mode not in _MODES:
            mode = None          # (the reference ignores an unknown mode)
        for stage in _STAGES:
            values = list(schedule[stage])
            if stage in _MODES[mode]:
                values[0] = _MODES[mode][stage]
            for key, value in zip(_SCHEDULE_KEYS, values):
                setattr(self, "%s_%s" % (stage, key), value)
        if not np.allclose(awin, awin[::-1]):
            print('WARNING: It appears you are using an analysis window that is not symmetric.\n'
                  'The current code uses simplifications that rely on such symmetry, so the code may not behave properly.')
        self.device = int(device)
        self._plan_kw = dict(device=device, precision=precision, nofuture_q4_compat=nofuture_q4_compat,
                             force_generic=force_generic, storage=storage)
        self._plan = None

    # -- engine plumbing (not part of the reference surface) --
    def plan(self):
        """The cached device plan holding W, W_ai, W_af for this configuration."""
        if self._plan is None:
            self._plan = _capi.Plan(self.fsize // 2 + 1, self.W, self.W_ai, self.W_af, **self._plan_kw)
        return self._plan

    def _qdiv(self):
        return float(self.fsize / self.fshift)

    def _as_c128(self, S):
        S = np.asarray(S)
        return S if S.dtype == np.complex128 else S.astype(np.complex128)

    def _check(self, S):
        F = S.shape[-1]
        if F % 2 == 0:
            raise ValueError('Please only include non-negative frequencies in the input spectrogram.')
        if F != self.fsize // 2 + 1:
            raise ValueError('This lws object works on %d-bin spectrograms, got %d.' % (self.fsize // 2 + 1, F))

    # -- reference surface --
    def get_consistency(self, S):
        return get_consistency(S, self.fsize, self.fshift, self.awin, self.swin, perfectrec=self.perfectrec)

    def stft(self, S):
        return stft(S, self.fsize, self.fshift, self.awin, perfectrec=self.perfectrec)

    # ---- device versions of the three helpers above (float32 transforms; any even frame size in 32..4096).
    # Arguments are torch CUDA tensors (complex64 spectrograms (B, T, F) / float32 signals (B, len)) or numpy arrays,
    # which are moved to the plan's device through torch -- PyTorch is only the owner of the device memory here.
    def _to_dev(self, a, dtype):
        import torch
        dev = torch.device("cuda", self.device)
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device=dev, dtype=dtype).contiguous()

    def get_consistency_dev(self, S):
        """Consistency in dB of each spectrogram of the stack S (B, T, F) or of the single spectrogram (T, F), computed
        on the device (lws.pyx:140-144)."""
        import torch
        t = self._to_dev(S, torch.complex64)
        single = t.dim() == 2
        if single:
            t = t[None]
        sums = _capi.consistency_dev(t.data_ptr(), t.shape[0], t.shape[1], self.fsize, self.fshift, self.awin, self.swin,
                                     self.perfectrec, device=self.device,
                                     stream=torch.cuda.current_stream(t.device).cuda_stream)
        db = 10 * np.log10(sums[:, 0] / sums[:, 1])
        return float(db[0]) if single else db

    def stft_dev(self, x):
        """STFT of the signals x (B, len) (or one signal (len,)) on the device: complex64 torch tensor (B, T, F)."""
        import torch
        t = self._to_dev(x, torch.float32)
        single = t.dim() == 1
        if single:
            t = t[None]
        B, n = t.shape
        T = _capi.stft_frames(n, self.fsize, self.fshift, self.perfectrec)
        out = torch.empty((B, T, self.fsize // 2 + 1), dtype=torch.complex64, device=t.device)
        _capi.stft_dev(t.data_ptr(), B, n, self.fsize, self.fshift, self.awin, self.perfectrec, out.data_ptr(),
                       device=self.device, stream=torch.cuda.current_stream(t.device).cuda_stream)
        return out[0] if single else out

    def istft_dev(self, S):
        """Inverse STFT of the spectrograms S (B, T, F) (or one (T, F)) on the device: float32 torch tensor (B, len)."""
        import torch
        t = self._to_dev(S, torch.complex64)
        single = t.dim() == 2
        if single:
            t = t[None]
        B, T, _ = t.shape
        n = _capi.istft_length(T, self.fsize, self.fshift, self.perfectrec)
        out = torch.empty((B, n), dtype=torch.float32, device=t.device)
        _capi.istft_dev(t.data_ptr(), B, T, self.fsize, self.fshift, self.swin, self.perfectrec, out.data_ptr(),
                        device=self.device, stream=torch.cuda.current_stream(t.device).cuda_stream)
        return out[0] if single else out

    def istft(self, S):
        return istft(S, self.fshift, self.swin, perfectrec=self.perfectrec)

    def nofuture_lws(self, S, iterations=None, thresholds=None):
        if iterations is None:
            iterations = self.nofuture_iterations
        if thresholds is None:
            thresholds = get_thresholds(iterations, self.nofuture_alpha, self.nofuture_beta, self.nofuture_gamma)
        S = self._as_c128(S)
        if len(thresholds) == 0:
            return S
        self._check(S)
        return self.plan().nofuture(S, thresholds, wsel=_capi.LWS_W_AI)  # W_ai on purpose: lws.pyx:475

    def online_lws(self, S, iterations=None, thresholds=None):
        if iterations is None:
            iterations = self.online_iterations
        if thresholds is None:
            thresholds = get_thresholds(iterations, self.online_alpha, self.online_beta, self.online_gamma)
        S = self._as_c128(S)
        if len(thresholds) == 0:
            return S
        self._check(S)
        return self.plan().online(S, thresholds, self.look_ahead, self._qdiv())

    def batch_lws(self, S, iterations=None, thresholds=None):
        if iterations is None:
            iterations = self.batch_iterations
        if thresholds is None:
            thresholds = get_thresholds(iterations, self.batch_alpha, self.batch_beta, self.batch_gamma)
        S = self._as_c128(S)
        if len(thresholds) == 0:
            return S
        self._check(S)
        return self.plan().batch(S, thresholds)

    def run_lws(self, S):
        """nofuture -> online -> batch (lws.pyx:495-499) as one device-resident pipeline."""
        S = self._as_c128(S)
        t0 = get_thresholds(self.nofuture_iterations, self.nofuture_alpha, self.nofuture_beta, self.nofuture_gamma)
        t1 = get_thresholds(self.online_iterations, self.online_alpha, self.online_beta, self.online_gamma)
        t2 = get_thresholds(self.batch_iterations, self.batch_alpha, self.batch_beta, self.batch_gamma)
        if len(t0) + len(t1) + len(t2) == 0:
            return S
        self._check(S)
        return self.plan().run(S, t0, t1, self.look_ahead, self._qdiv(), t2)
