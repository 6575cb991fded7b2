"""Python binding of liblws_hip.so (include/lws_hip.h).

Two interchangeable bindings of the same C ABI:
  * ``lws_amd/_cylws`` -- the Cython shim (``_cylws.pyx`` over ``csrc/lws_hip.pxd``), the counterpart of the reference's
    python/lws.pyx over python/lwslib.pxd:1-13; every call releases the GIL.  Built by ``make -C lws_amd/csrc``.
  * ctypes -- the same prototypes declared below; used when the extension is absent (or ``LWS_BINDING=ctypes``).
``load()`` returns whichever is in use; both expose the C functions under their C names with the C argument order.
The library itself is mandatory: if it cannot be loaded the LWS entry points raise -- there is no CPU fallback in the
product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LWS_HIP_LIB", os.path.join(_HERE, "liblws_hip.so"))  # override: kernel-variant experiments

# lws_hip.h enums
LWS_OK, LWS_ERR_INVALID, LWS_ERR_HIP, LWS_ERR_NOMEM, LWS_ERR_UNSUPPORTED = range(5)
LWS_PRECISION_FP32, LWS_PRECISION_FP64, LWS_NOFUTURE_Q4_COMPAT, LWS_FORCE_GENERIC, LWS_NO_DIRECT_IO = 0, 1, 2, 4, 8
LWS_STORAGE_FP16 = 16
LWS_GENERIC_PLAIN_LAYOUT = 32
LWS_W, LWS_W_AI, LWS_W_AF = 0, 1, 2

EXPORTS = (
    "lws_hip_version", "lws_last_error", "lws_device_count", "lws_plan_create", "lws_plan_destroy",
    "lws_batch_lws", "lws_nofuture_lws", "lws_online_lws", "lws_run_lws", "lws_batch_lws_dev",
    "lws_nofuture_lws_dev", "lws_online_lws_dev", "lws_residual_dev", "lws_last_kernel_time",
    "lws_last_kernel_name", "lws_generic_stage", "lws_stft_frames", "lws_istft_length", "lws_stft_dev", "lws_istft_dev", "lws_stft_zp_dev",
    "lws_consistency_dev", "lws_hann", "lws_synthwin", "lws_weights_shape", "lws_create_weights",
    "lws_build_asymmetric_windows", "lws_get_thresholds", "lws_plan_create_from_windows", "lws_stream_copy",
    "lws_run_lws_dev", "lws_plan_reserve", "lws_residual", "lws_residual_allreduce_dev", "lws_weights_structure", "lws_multi_plan_create", "lws_multi_plan_destroy",
    "lws_multi_plan_shards", "lws_multi_batch_lws", "lws_multi_run_lws", "lws_multi_residual",
)

_lib = None
BINDING = None   # "cython" or "ctypes", set by load()


class LwsHipError(RuntimeError):
    """A non-OK status from liblws_hip.so that is not an argument error."""


def load():
    """Load liblws_hip.so once and declare its prototypes.  Raises OSError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lws_amd/csrc` (hipcc, gfx950).  lws_amd has no CPU fallback.")
    # PyTorch's ROCm wheels bundle their own HIP/HSA runtime next to the system one this library links.  Both can
    # live in one process only if PyTorch's is initialised first (the other order leaves torch with "No HIP GPUs are
    # available"), so do that here when PyTorch is installed -- the *_dev entry points are meant to be fed from it.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # no torch, or no usable device: nothing to order
        pass
    global BINDING
    if os.environ.get("LWS_BINDING", "cython") != "ctypes" and "LWS_HIP_LIB" not in os.environ:
        try:
            from . import _cylws   # links liblws_hip.so next to it (rpath $ORIGIN)
            for name in EXPORTS:
                getattr(_cylws, name)
            _lib, BINDING = _cylws, "cython"
            return _lib
        except ImportError:
            pass   # extension not built: the ctypes binding below serves the same ABI
    lib = C.CDLL(LIB_PATH)
    BINDING = "ctypes"
    vp, ip = C.c_void_p, C.c_int
    lib.lws_hip_version.restype = C.c_int
    lib.lws_last_error.restype = C.c_char_p
    lib.lws_device_count.restype = C.c_int
    lib.lws_plan_create.argtypes = [C.POINTER(vp), ip, ip, ip, ip, ip, vp, vp, vp, C.c_uint]
    lib.lws_plan_destroy.argtypes = [vp]
    lib.lws_plan_destroy.restype = None
    lib.lws_batch_lws.argtypes = [vp, ip, vp, vp, ip, ip, vp, ip]
    lib.lws_nofuture_lws.argtypes = [vp, ip, vp, vp, ip, ip, vp, ip]
    lib.lws_online_lws.argtypes = [vp, vp, vp, ip, ip, vp, ip, ip, C.c_double]
    lib.lws_run_lws.argtypes = [vp, vp, vp, ip, ip, vp, ip, vp, ip, ip, C.c_double, vp, ip]
    lib.lws_batch_lws_dev.argtypes = [vp, ip, vp, ip, ip, vp, ip, vp]
    lib.lws_nofuture_lws_dev.argtypes = [vp, ip, vp, ip, ip, vp, ip, vp]
    lib.lws_online_lws_dev.argtypes = [vp, vp, ip, ip, vp, ip, ip, C.c_double, vp]
    lib.lws_residual_dev.argtypes = [vp, vp, ip, ip, vp, vp]
    lib.lws_last_kernel_time.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.lws_last_kernel_name.argtypes = [vp]
    lib.lws_stream_copy.argtypes = [vp, vp, C.c_size_t, vp]
    lib.lws_last_kernel_name.restype = C.c_char_p
    lib.lws_generic_stage.argtypes = [vp]
    lib.lws_generic_stage.restype = C.c_char_p
    lib.lws_stft_frames.argtypes = [ip, ip, ip, ip]
    lib.lws_istft_length.argtypes = [ip, ip, ip, ip]
    lib.lws_stft_dev.argtypes = [ip, vp, ip, ip, ip, ip, vp, ip, vp, vp]
    lib.lws_istft_dev.argtypes = [ip, vp, ip, ip, ip, ip, vp, ip, vp, vp]
    lib.lws_stft_zp_dev.argtypes = [ip, vp, ip, ip, ip, ip, ip, vp, ip, vp, vp]
    lib.lws_consistency_dev.argtypes = [ip, vp, ip, ip, ip, ip, vp, vp, ip, vp, vp]
    dp = C.c_double
    lib.lws_hann.argtypes = [ip, ip, ip, vp]
    lib.lws_synthwin.argtypes = [vp, ip, ip, vp, vp]
    lib.lws_weights_shape.argtypes = [ip, ip, ip, C.POINTER(ip), C.POINTER(ip)]
    lib.lws_create_weights.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    lib.lws_build_asymmetric_windows.argtypes = [vp, ip, ip, vp, vp]
    lib.lws_get_thresholds.argtypes = [ip, dp, dp, dp, vp]
    lib.lws_plan_create_from_windows.argtypes = [C.POINTER(vp), ip, vp, vp, ip, ip, ip, ip, C.c_uint, vp, vp]
    lib.lws_run_lws_dev.argtypes = [vp, vp, ip, ip, vp, ip, vp, ip, ip, C.c_double, vp, ip, vp]
    lib.lws_plan_reserve.argtypes = [vp, ip, ip, ip]
    lib.lws_residual.argtypes = [vp, vp, ip, ip, vp]
    lib.lws_residual_allreduce_dev.argtypes = [vp, vp, ip, ip, vp, vp, vp]
    lib.lws_weights_structure.argtypes = [vp, ip, ip, ip, vp, vp]
    lib.lws_multi_plan_create.argtypes = [C.POINTER(vp), ip, vp, ip, ip, ip, ip, vp, vp, vp, C.c_uint]
    lib.lws_multi_plan_destroy.argtypes = [vp]
    lib.lws_multi_plan_destroy.restype = None
    lib.lws_multi_plan_shards.argtypes = [vp]
    lib.lws_multi_batch_lws.argtypes = [vp, ip, vp, vp, ip, ip, vp, ip]
    lib.lws_multi_run_lws.argtypes = [vp, vp, vp, ip, ip, vp, ip, vp, ip, ip, C.c_double, vp, ip]
    lib.lws_multi_residual.argtypes = [vp, vp, ip, ip, vp]
    for name in EXPORTS:  # fail at load time, not at first use, if a symbol is missing
        getattr(lib, name)
    _lib = lib
    return lib


def load_raw():
    """The library as a plain ctypes handle (no prototypes declared): for symbols outside include/lws_hip.h, i.e. the
    C++-mangled lwslib.h interface of include/lwslib_compat.h."""
    load()   # (initialisation order with PyTorch's runtime, see load())
    return C.CDLL(LIB_PATH)


def check(rc):
    """Map a status code to the exception the reference's Python layer would raise."""
    if rc == LWS_OK:
        return
    msg = load().lws_last_error().decode("utf-8", "replace")
    if rc == LWS_ERR_INVALID:
        raise ValueError(msg)  # lws.pyx:224,277,337 raise ValueError for bad shapes
    if rc == LWS_ERR_NOMEM:
        raise MemoryError(msg)
    raise LwsHipError(f"liblws_hip status {rc}: {msg}")


def _c128(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    return a


class Plan:
    """Owns an ``lws_plan`` (device copies of W / W_ai / W_af for one (F, L, Q) shape)."""

    def __init__(self, F, W, W_ai=None, W_af=None, device=0, precision="fp32",
                 nofuture_q4_compat=True, force_generic=False, direct_io=True, storage="fp32", generic_plain_layout=False):
        lib = load()
        W = _c128(W)
        if W.ndim != 3:
            raise ValueError("weights must have shape (Qprime, Q, L+1)")
        self.Qp, self.Q, self.L = W.shape[0], W.shape[1], W.shape[2] - 1
        self.F = int(F)
        self._keep = [W]
        ptrs = [W.ctypes.data]
        for other in (W_ai, W_af):
            if other is None:
                ptrs.append(None)
                continue
            other = _c128(other)
            if other.shape != W.shape:
                raise ValueError("W, W_ai and W_af must have the same shape")
            self._keep.append(other)
            ptrs.append(other.ctypes.data)
        flags = 0
        if precision == "fp64":
            flags |= LWS_PRECISION_FP64
        elif precision != "fp32":
            raise ValueError("precision must be 'fp32' or 'fp64'")
        if nofuture_q4_compat:
            flags |= LWS_NOFUTURE_Q4_COMPAT
        if force_generic:
            flags |= LWS_FORCE_GENERIC
        if not direct_io:
            flags |= LWS_NO_DIRECT_IO
        if generic_plain_layout:
            flags |= LWS_GENERIC_PLAIN_LAYOUT
        if storage == "fp16":
            flags |= LWS_STORAGE_FP16
        elif storage != "fp32":
            raise ValueError("storage must be 'fp32' or 'fp16'")
        self.storage = storage
        self.precision = precision
        self.device = int(device)
        h = C.c_void_p()
        check(lib.lws_plan_create(C.byref(h), self.device, self.F, self.L, self.Q, self.Qp,
                                  ptrs[0], ptrs[1], ptrs[2], flags))
        self._h = h
        self._lib = lib
        self._expect_generic = bool(force_generic) or precision == "fp64"
        self._warned = False

    def _note_engine(self):
        """The generic engine serves every shape and weight tensor the reference accepts, 20-40x slower than the kernels
        built for the common ones: say so once per plan instead of leaving lws_last_kernel_name as the only tell."""
        if self._warned or self._expect_generic:
            return
        name = self._lib.lws_last_kernel_name(self._h).decode()      # (no synchronisation: a string set at launch time)
        stage = self._lib.lws_generic_stage(self._h).decode()        # any stage of a pipeline, not just the last one
        if name.startswith("generic") or stage:
            if not name.startswith("generic"):
                name = "the %s stage" % stage
            self._warned = True
            import warnings
            warnings.warn(
                "lws_amd: this plan (F=%d bins, Q=%d, L=%d%s) runs on the order-exact generic engine (%s), 20-40x slower than "
                "the systolic / band / LDS kernels; those serve batch sweeps of plans with create_weights() tensors (summarised, or general "
                "with rows that repeat) of up to 16 frames per stencil row, L <= 10 and frames of 17 bins or more (band engine: as long as one "
                "sweep slot's ring fits the LDS); online and no-future sweeps of any tensor run on their LDS engines or on the team engine "
                "unless a wavefront step holds more than 512 bins (long frames at a small Q)"
                % (self.F, self.Q, self.L, "" if self.Qp == self.Q else ", general weights", name), RuntimeWarning, stacklevel=3)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lws_plan_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    # ---- host (numpy complex128) entry points ----
    def _io(self, S):
        S = _c128(S)
        if S.ndim == 2:
            S3 = S[None]
        elif S.ndim == 3:
            S3 = S
        else:
            raise ValueError("expected a (T, F) spectrogram or a (B, T, F) stack")
        if S3.shape[2] != self.F:
            raise ValueError(f"plan was built for F={self.F} bins, got {S3.shape[2]}")
        out = np.empty_like(S3)
        return S, S3, out

    @staticmethod
    def _thr(thresholds):
        t = np.ascontiguousarray(thresholds, dtype=np.float64).ravel()
        return t, (t.ctypes.data if t.size else None)

    def batch(self, S, thresholds, wsel=LWS_W):
        S, S3, out = self._io(S)
        t, tp = self._thr(thresholds)
        check(self._lib.lws_batch_lws(self._h, wsel, S3.ctypes.data, out.ctypes.data, S3.shape[0],
                                      S3.shape[1], tp, t.size))
        self._note_engine()
        return out.reshape(S.shape)

    def nofuture(self, S, thresholds, wsel=LWS_W):
        S, S3, out = self._io(S)
        t, tp = self._thr(thresholds)
        check(self._lib.lws_nofuture_lws(self._h, wsel, S3.ctypes.data, out.ctypes.data,
                                         S3.shape[0], S3.shape[1], tp, t.size))
        self._note_engine()
        return out.reshape(S.shape)

    def online(self, S, thresholds, LA, qdiv):
        S, S3, out = self._io(S)
        t, tp = self._thr(thresholds)
        check(self._lib.lws_online_lws(self._h, S3.ctypes.data, out.ctypes.data, S3.shape[0],
                                       S3.shape[1], tp, t.size, int(LA), float(qdiv)))
        self._note_engine()
        return out.reshape(S.shape)

    def run(self, S, thr_nofuture, thr_online, LA, qdiv, thr_batch):
        S, S3, out = self._io(S)
        t0, p0 = self._thr(thr_nofuture)
        t1, p1 = self._thr(thr_online)
        t2, p2 = self._thr(thr_batch)
        check(self._lib.lws_run_lws(self._h, S3.ctypes.data, out.ctypes.data, S3.shape[0], S3.shape[1],
                                    p0, t0.size, p1, t1.size, int(LA), float(qdiv), p2, t2.size))
        self._note_engine()
        return out.reshape(S.shape)

    # ---- device-resident entry points (raw pointers: torch tensors pass .data_ptr()) ----
    def batch_dev(self, ptr, B, T, thresholds, wsel=LWS_W, stream=None):
        t, tp = self._thr(thresholds)
        check(self._lib.lws_batch_lws_dev(self._h, wsel, ptr, B, T, tp, t.size, stream))
        self._note_engine()

    def nofuture_dev(self, ptr, B, T, thresholds, wsel=LWS_W, stream=None):
        t, tp = self._thr(thresholds)
        check(self._lib.lws_nofuture_lws_dev(self._h, wsel, ptr, B, T, tp, t.size, stream))
        self._note_engine()

    def online_dev(self, ptr, B, T, thresholds, LA, qdiv, stream=None):
        t, tp = self._thr(thresholds)
        check(self._lib.lws_online_lws_dev(self._h, ptr, B, T, tp, t.size, int(LA), float(qdiv), stream))
        self._note_engine()

    def run_dev(self, ptr, B, T, thr_nofuture, thr_online, LA, qdiv, thr_batch, stream=None):
        t0, p0 = self._thr(thr_nofuture)
        t1, p1 = self._thr(thr_online)
        t2, p2 = self._thr(thr_batch)
        check(self._lib.lws_run_lws_dev(self._h, ptr, B, T, p0, t0.size, p1, t1.size, int(LA), float(qdiv), p2, t2.size, stream))
        self._note_engine()

    def reserve(self, B, T, max_iters):
        """Pre-size all scratch so that later *_dev calls of up to this shape only enqueue work (no hipMalloc)."""
        check(self._lib.lws_plan_reserve(self._h, int(B), int(T), int(max_iters)))

    def residual(self, S):
        S, S3, _ = self._io(S)
        out = np.empty((S3.shape[0], 2), dtype=np.float64)
        check(self._lib.lws_residual(self._h, S3.ctypes.data, S3.shape[0], S3.shape[1], out.ctypes.data))
        return out

    def residual_allreduce_dev(self, ptr, B, T, comm=None, stream=None):
        """[sum |acc + w00 S|^2, sum |S|^2] over this rank's B device spectrograms, all-reduced over the ranks of the RCCL
        communicator `comm` (an ncclComm_t as an integer; None: this rank's sums)."""
        out = np.empty(2, dtype=np.float64)
        check(self._lib.lws_residual_allreduce_dev(self._h, ptr, B, T, comm, out.ctypes.data, stream))
        return out

    def residual_dev(self, ptr, B, T, stream=None):
        out = np.empty((B, 2), dtype=np.float64)
        check(self._lib.lws_residual_dev(self._h, ptr, B, T, out.ctypes.data, stream))
        return out

    def last_kernel(self):
        ms, n = C.c_float(), C.c_int()
        check(self._lib.lws_last_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return {"ms": ms.value, "launches": n.value,
                "name": self._lib.lws_last_kernel_name(self._h).decode()}


class MultiPlan:
    """One plan per device behind one handle (``lws_multi_*``): a batch is dealt in contiguous blocks to the devices,
    one host thread per device inside the library.  ``devices``: list of HIP ordinals (a device may repeat), or None
    for every visible device."""

    def __init__(self, F, W, W_ai=None, W_af=None, devices=None, precision="fp32", nofuture_q4_compat=True,
                 force_generic=False, storage="fp32"):
        lib = load()
        W = _c128(W)
        self.Qp, self.Q, self.L = W.shape[0], W.shape[1], W.shape[2] - 1
        self.F = int(F)
        self._keep = [W]
        ptrs = [W.ctypes.data]
        for other in (W_ai, W_af):
            if other is None:
                ptrs.append(None)
                continue
            other = _c128(other)
            self._keep.append(other)
            ptrs.append(other.ctypes.data)
        flags = (LWS_PRECISION_FP64 if precision == "fp64" else 0) | (LWS_NOFUTURE_Q4_COMPAT if nofuture_q4_compat else 0) \
            | (LWS_FORCE_GENERIC if force_generic else 0) | (LWS_STORAGE_FP16 if storage == "fp16" else 0)
        devs = None if devices is None else np.ascontiguousarray(devices, dtype=np.intc)
        h = C.c_void_p()
        check(lib.lws_multi_plan_create(C.byref(h), 0 if devs is None else int(devs.size), None if devs is None else devs.ctypes.data,
                                        self.F, self.L, self.Q, self.Qp, ptrs[0], ptrs[1], ptrs[2], flags))
        self._h, self._lib = h, lib
        self.shards = lib.lws_multi_plan_shards(h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lws_multi_plan_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _io(self, S):
        S = _c128(S)
        if S.ndim != 3 or S.shape[2] != self.F:
            raise ValueError(f"expected a (B, T, {self.F}) stack of spectrograms")
        return S, np.empty_like(S)

    def batch(self, S, thresholds, wsel=LWS_W):
        S, out = self._io(S)
        t, tp = Plan._thr(thresholds)
        check(self._lib.lws_multi_batch_lws(self._h, wsel, S.ctypes.data, out.ctypes.data, S.shape[0], S.shape[1], tp, t.size))
        return out

    def run(self, S, thr_nofuture, thr_online, LA, qdiv, thr_batch):
        S, out = self._io(S)
        t0, p0 = Plan._thr(thr_nofuture)
        t1, p1 = Plan._thr(thr_online)
        t2, p2 = Plan._thr(thr_batch)
        check(self._lib.lws_multi_run_lws(self._h, S.ctypes.data, out.ctypes.data, S.shape[0], S.shape[1], p0, t0.size, p1, t1.size,
                                          int(LA), float(qdiv), p2, t2.size))
        return out

    def residual(self, S):
        S, _ = self._io(S)
        out = np.empty(2, dtype=np.float64)
        check(self._lib.lws_multi_residual(self._h, S.ctypes.data, S.shape[0], S.shape[1], out.ctypes.data))
        return out


# ---- the steps either side of the path, on the device (include/lws_hip.h; lws.pyx:43-144) --------------------------
def _win(w):
    return np.ascontiguousarray(w, dtype=np.float64)


def stft_frames(length, fsize, fshift, perfectrec):
    return load().lws_stft_frames(int(length), int(fsize), int(fshift), int(bool(perfectrec)))


def istft_length(frames, fsize, fshift, perfectrec):
    return load().lws_istft_length(int(frames), int(fsize), int(fshift), int(bool(perfectrec)))


def stft_dev(x_ptr, B, length, fsize, fshift, awin, perfectrec, S_ptr, device=0, stream=None, fftsize=None):
    """x_ptr: device float32 [B][length]; S_ptr: device complex64 [B][stft_frames(...)][fftsize//2+1] (fftsize: fsize unless given)."""
    a = _win(awin)
    if int(fsize) % 2:
        # (the reference's stft only asks for an even fftsize, lws.pyx:49-50: an odd frame under an even, longer transform works on
        #  the host path; the device kernels pair up the samples of a frame)
        raise ValueError("stft_dev needs an even frame size (got %d): use the host stft() for an odd frame under a longer transform" % int(fsize))
    if fftsize is None or int(fftsize) == int(fsize):
        check(load().lws_stft_dev(int(device), x_ptr, int(B), int(length), int(fsize), int(fshift), a.ctypes.data,
                                  int(bool(perfectrec)), S_ptr, stream))
    else:
        check(load().lws_stft_zp_dev(int(device), x_ptr, int(B), int(length), int(fsize), int(fftsize), int(fshift), a.ctypes.data,
                                     int(bool(perfectrec)), S_ptr, stream))


def istft_dev(S_ptr, B, frames, fsize, fshift, swin, perfectrec, x_ptr, device=0, stream=None):
    """S_ptr: device complex64 [B][frames][fsize//2+1]; x_ptr: device float32 [B][istft_length(...)]."""
    w = _win(swin)
    check(load().lws_istft_dev(int(device), S_ptr, int(B), int(frames), int(fsize), int(fshift), w.ctypes.data,
                               int(bool(perfectrec)), x_ptr, stream))


def weights_structure(W):
    """(period, step) of the twiddle structure of a weight tensor W[Qp][Q][L+1] -- W[p][r][k] == W[0][r][k] exp(2j pi p r step / period)
    for every row p; create_weights gives period / step = frame / hop in lowest terms -- or None if it has none (such plans run on the
    generic engine).  Host-only: no device needed."""
    W = np.ascontiguousarray(W, dtype=np.complex128)
    per, stp = np.zeros(1, dtype=np.intc), np.zeros(1, dtype=np.intc)
    ok = load().lws_weights_structure(W.ctypes.data, int(W.shape[1]), int(W.shape[0]), int(W.shape[2]) - 1, per.ctypes.data, stp.ctypes.data)
    return (int(per[0]), int(stp[0])) if ok else None


def consistency_dev(S_ptr, B, frames, fsize, fshift, awin, swin, perfectrec, device=0, stream=None):
    """Per spectrogram [sum |S|^2, sum |stft(istft(S)) - S|^2] (fp64, host array (B, 2)); dB = 10 log10(ratio)."""
    a, w = _win(awin), _win(swin)
    out = np.empty((int(B), 2), dtype=np.float64)
    check(load().lws_consistency_dev(int(device), S_ptr, int(B), int(frames), int(fsize), int(fshift), a.ctypes.data,
                                     w.ctypes.data, int(bool(perfectrec)), out.ctypes.data, stream))
    return out
