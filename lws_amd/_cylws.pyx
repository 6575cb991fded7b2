# cython: language_level=3
"""_cylws -- the Cython shim over the C ABI of liblws_hip.so (include/lws_hip.h).

The reference binds its C++ kernels with Cython (python/lws.pyx over python/lwslib.pxd:1-13, built by
python/setup.py:71-75); this is the same kind of binding over the new boundary (lws_amd/csrc/lws_hip.pxd).  Every
call into the library runs ``with nogil`` -- the reference holds the GIL for a whole call -- so several Python
threads can drive several plans / devices.

Functions carry the C names and the C argument order, and take pointers the way ``lws_amd._capi`` already passes
them to ctypes (an address as int, None for NULL, a ctypes ``c_void_p`` or ``byref(...)`` object), so
``_capi.load()`` can hand out this module in place of the ctypes library object: same call sites, either binding.
"""
import ctypes as _C
from libc.stdint cimport uintptr_t
from libc.stddef cimport size_t
cimport lws_hip as c

BINDING = "cython"


cdef uintptr_t _addr(object x) except? 1:
    """Address carried by x: None -> NULL, int, ctypes pointer-like (.value), ctypes byref() (._obj)."""
    if x is None:
        return 0
    if isinstance(x, int):
        return <uintptr_t>x
    obj = getattr(x, "_obj", None)
    if obj is not None:                       # ctypes.byref(obj)
        return <uintptr_t>_C.addressof(obj)
    v = getattr(x, "value", x)
    if v is None:
        return 0
    return <uintptr_t>int(v)


cdef double _dbl(object x) except? -1.0e300:
    """A C double from a Python number or a ctypes c_double."""
    return float(getattr(x, "value", x))


def lws_hip_version():
    return c.lws_hip_version()


def lws_last_error():
    cdef const char *s = c.lws_last_error()
    return <bytes>s if s != NULL else b""


def lws_device_count():
    cdef int n
    with nogil:
        n = c.lws_device_count()
    return n


def lws_plan_create(plan_ref, int device, int F, int L, int Q, int Qp, W, W_ai, W_af, unsigned flags):
    cdef c.lws_plan *p = NULL
    cdef uintptr_t w = _addr(W), wai = _addr(W_ai), waf = _addr(W_af)
    cdef int rc
    with nogil:
        rc = c.lws_plan_create(&p, device, F, L, Q, Qp, <const double *>w, <const double *>wai, <const double *>waf, flags)
    plan_ref._obj.value = <uintptr_t>p if p != NULL else None
    return rc


def lws_plan_create_from_windows(plan_ref, int device, awin, swin, int fsize, int fshift, int L, int symmetric_win,
                                 unsigned flags, awin_out, swin_out):
    cdef c.lws_plan *p = NULL
    cdef uintptr_t a = _addr(awin), s = _addr(swin), ao = _addr(awin_out), so = _addr(swin_out)
    cdef int rc
    with nogil:
        rc = c.lws_plan_create_from_windows(&p, device, <const double *>a, <const double *>s, fsize, fshift, L, symmetric_win,
                                            flags, <double *>ao, <double *>so)
    plan_ref._obj.value = <uintptr_t>p if p != NULL else None
    return rc


def lws_plan_destroy(plan):
    cdef uintptr_t p = _addr(plan)
    with nogil:
        c.lws_plan_destroy(<c.lws_plan *>p)


def lws_plan_reserve(plan, int B, int T, int max_iters):
    cdef uintptr_t p = _addr(plan)
    cdef int rc
    with nogil:
        rc = c.lws_plan_reserve(<c.lws_plan *>p, B, T, max_iters)
    return rc


def lws_batch_lws(plan, int wsel, S_in, S_out, int B, int T, thresholds, int iters):
    cdef uintptr_t p = _addr(plan), i = _addr(S_in), o = _addr(S_out), t = _addr(thresholds)
    cdef int rc
    with nogil:
        rc = c.lws_batch_lws(<c.lws_plan *>p, wsel, <const double *>i, <double *>o, B, T, <const double *>t, iters)
    return rc


def lws_nofuture_lws(plan, int wsel, S_in, S_out, int B, int T, thresholds, int iters):
    cdef uintptr_t p = _addr(plan), i = _addr(S_in), o = _addr(S_out), t = _addr(thresholds)
    cdef int rc
    with nogil:
        rc = c.lws_nofuture_lws(<c.lws_plan *>p, wsel, <const double *>i, <double *>o, B, T, <const double *>t, iters)
    return rc


def lws_online_lws(plan, S_in, S_out, int B, int T, thresholds, int iters, int LA, qdiv_):
    cdef uintptr_t p = _addr(plan), i = _addr(S_in), o = _addr(S_out), t = _addr(thresholds)
    cdef double qdiv = _dbl(qdiv_)
    cdef int rc
    with nogil:
        rc = c.lws_online_lws(<c.lws_plan *>p, <const double *>i, <double *>o, B, T, <const double *>t, iters, LA, qdiv)
    return rc


def lws_run_lws(plan, S_in, S_out, int B, int T, thr_nofuture, int it_nofuture, thr_online, int it_online, int LA,
                qdiv_, thr_batch, int it_batch):
    cdef uintptr_t p = _addr(plan), i = _addr(S_in), o = _addr(S_out)
    cdef double qdiv = _dbl(qdiv_)
    cdef uintptr_t t0 = _addr(thr_nofuture), t1 = _addr(thr_online), t2 = _addr(thr_batch)
    cdef int rc
    with nogil:
        rc = c.lws_run_lws(<c.lws_plan *>p, <const double *>i, <double *>o, B, T, <const double *>t0, it_nofuture,
                           <const double *>t1, it_online, LA, qdiv, <const double *>t2, it_batch)
    return rc


def lws_batch_lws_dev(plan, int wsel, S_dev, int B, int T, thresholds, int iters, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), t = _addr(thresholds), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_batch_lws_dev(<c.lws_plan *>p, wsel, <void *>s, B, T, <const double *>t, iters, <void *>st)
    return rc


def lws_nofuture_lws_dev(plan, int wsel, S_dev, int B, int T, thresholds, int iters, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), t = _addr(thresholds), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_nofuture_lws_dev(<c.lws_plan *>p, wsel, <void *>s, B, T, <const double *>t, iters, <void *>st)
    return rc


def lws_online_lws_dev(plan, S_dev, int B, int T, thresholds, int iters, int LA, qdiv_, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), t = _addr(thresholds), st = _addr(stream)
    cdef double qdiv = _dbl(qdiv_)
    cdef int rc
    with nogil:
        rc = c.lws_online_lws_dev(<c.lws_plan *>p, <void *>s, B, T, <const double *>t, iters, LA, qdiv, <void *>st)
    return rc


def lws_run_lws_dev(plan, S_dev, int B, int T, thr_nofuture, int it_nofuture, thr_online, int it_online, int LA,
                    qdiv_, thr_batch, int it_batch, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), st = _addr(stream)
    cdef double qdiv = _dbl(qdiv_)
    cdef uintptr_t t0 = _addr(thr_nofuture), t1 = _addr(thr_online), t2 = _addr(thr_batch)
    cdef int rc
    with nogil:
        rc = c.lws_run_lws_dev(<c.lws_plan *>p, <void *>s, B, T, <const double *>t0, it_nofuture, <const double *>t1,
                               it_online, LA, qdiv, <const double *>t2, it_batch, <void *>st)
    return rc


def lws_residual_dev(plan, S_dev, int B, int T, out, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), o = _addr(out), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_residual_dev(<c.lws_plan *>p, <const void *>s, B, T, <double *>o, <void *>st)
    return rc


def lws_weights_structure(W, int Q, int Qp, int L, period, step):
    cdef uintptr_t w = _addr(W), pp = _addr(period), ss = _addr(step)
    return c.lws_weights_structure(<const double *>w, Q, Qp, L, <int *>pp, <int *>ss)


def lws_residual_allreduce_dev(plan, S_dev, int B, int T, comm, out, stream):
    cdef uintptr_t p = _addr(plan), s = _addr(S_dev), cm = _addr(comm), o = _addr(out), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_residual_allreduce_dev(<c.lws_plan *>p, <const void *>s, B, T, <void *>cm, <double *>o, <void *>st)
    return rc


def lws_residual(plan, S, int B, int T, out):
    cdef uintptr_t p = _addr(plan), s = _addr(S), o = _addr(out)
    cdef int rc
    with nogil:
        rc = c.lws_residual(<c.lws_plan *>p, <const double *>s, B, T, <double *>o)
    return rc


def lws_last_kernel_time(plan, ms_ref, launches_ref):
    cdef uintptr_t p = _addr(plan), m = _addr(ms_ref), n = _addr(launches_ref)
    cdef int rc
    with nogil:
        rc = c.lws_last_kernel_time(<c.lws_plan *>p, <float *>m, <int *>n)
    return rc


def lws_last_kernel_name(plan):
    cdef uintptr_t p = _addr(plan)
    cdef const char *s = c.lws_last_kernel_name(<c.lws_plan *>p)
    return <bytes>s if s != NULL else b"none"


def lws_generic_stage(plan):
    cdef uintptr_t p = _addr(plan)
    cdef const char *s = c.lws_generic_stage(<c.lws_plan *>p)
    return <bytes>s if s != NULL else b""


def lws_stream_copy(dst_dev, src_dev, size_t nbytes, stream):
    cdef uintptr_t d = _addr(dst_dev), s = _addr(src_dev), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_stream_copy(<void *>d, <const void *>s, nbytes, <void *>st)
    return rc


def lws_stft_frames(int length, int N, int fshift, int perfectrec):
    return c.lws_stft_frames(length, N, fshift, perfectrec)


def lws_istft_length(int M, int N, int fshift, int perfectrec):
    return c.lws_istft_length(M, N, fshift, perfectrec)


def lws_stft_dev(int device, x_dev, int B, int length, int N, int fshift, awin, int perfectrec, S_dev, stream):
    cdef uintptr_t x = _addr(x_dev), a = _addr(awin), s = _addr(S_dev), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_stft_dev(device, <const float *>x, B, length, N, fshift, <const double *>a, perfectrec, <void *>s, <void *>st)
    return rc


def lws_istft_dev(int device, S_dev, int B, int M, int N, int fshift, swin, int perfectrec, x_dev, stream):
    cdef uintptr_t x = _addr(x_dev), w = _addr(swin), s = _addr(S_dev), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_istft_dev(device, <const void *>s, B, M, N, fshift, <const double *>w, perfectrec, <float *>x, <void *>st)
    return rc


def lws_stft_zp_dev(int device, x_dev, int B, int length, int fsize, int fftsize, int fshift, awin, int perfectrec, S_dev, stream):
    cdef uintptr_t x = _addr(x_dev), a = _addr(awin), s = _addr(S_dev), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_stft_zp_dev(device, <const float *>x, B, length, fsize, fftsize, fshift, <const double *>a, perfectrec, <void *>s, <void *>st)
    return rc


def lws_consistency_dev(int device, S_dev, int B, int M, int N, int fshift, awin, swin, int perfectrec, out, stream):
    cdef uintptr_t a = _addr(awin), w = _addr(swin), s = _addr(S_dev), o = _addr(out), st = _addr(stream)
    cdef int rc
    with nogil:
        rc = c.lws_consistency_dev(device, <const void *>s, B, M, N, fshift, <const double *>a, <const double *>w, perfectrec,
                                   <double *>o, <void *>st)
    return rc


def lws_hann(int n, int symmetric, int use_offset, out):
    cdef uintptr_t o = _addr(out)
    return c.lws_hann(n, symmetric, use_offset, <double *>o)


def lws_synthwin(awin, int fsize, int fshift, swin, out):
    cdef uintptr_t a = _addr(awin), s = _addr(swin), o = _addr(out)
    return c.lws_synthwin(<const double *>a, fsize, fshift, <const double *>s, <double *>o)


def lws_weights_shape(int fsize, int fshift, int use_summarized_weights, qp_ref, q_ref):
    cdef uintptr_t qp = _addr(qp_ref), q = _addr(q_ref)
    return c.lws_weights_shape(fsize, fshift, use_summarized_weights, <int *>qp, <int *>q)


def lws_create_weights(awin, swin, int fsize, int fshift, int L, int use_summarized_weights, W):
    cdef uintptr_t a = _addr(awin), s = _addr(swin), w = _addr(W)
    return c.lws_create_weights(<const double *>a, <const double *>s, fsize, fshift, L, use_summarized_weights, <double *>w)


def lws_build_asymmetric_windows(awin_swin, int fsize, int fshift, win_ai, win_af):
    cdef uintptr_t a = _addr(awin_swin), i = _addr(win_ai), f = _addr(win_af)
    return c.lws_build_asymmetric_windows(<const double *>a, fsize, fshift, <double *>i, <double *>f)


def lws_get_thresholds(int iterations, alpha, beta, gamma, out):
    cdef uintptr_t o = _addr(out)
    return c.lws_get_thresholds(iterations, _dbl(alpha), _dbl(beta), _dbl(gamma), <double *>o)


# ---- one host thread per device (include/lws_hip.h, "multi-device") ----------------------------------------------------
def lws_multi_plan_create(mp_ref, int ndev, devices, int F, int L, int Q, int Qp, W, W_ai, W_af, unsigned flags):
    cdef c.lws_multi_plan *p = NULL
    cdef uintptr_t d = _addr(devices), w = _addr(W), wai = _addr(W_ai), waf = _addr(W_af)
    cdef int rc
    with nogil:
        rc = c.lws_multi_plan_create(&p, ndev, <const int *>d, F, L, Q, Qp, <const double *>w, <const double *>wai,
                                     <const double *>waf, flags)
    mp_ref._obj.value = <uintptr_t>p if p != NULL else None
    return rc


def lws_multi_plan_destroy(mp):
    cdef uintptr_t p = _addr(mp)
    with nogil:
        c.lws_multi_plan_destroy(<c.lws_multi_plan *>p)


def lws_multi_plan_shards(mp):
    cdef uintptr_t p = _addr(mp)
    return c.lws_multi_plan_shards(<c.lws_multi_plan *>p)


def lws_multi_batch_lws(mp, int wsel, S_in, S_out, int B, int T, thresholds, int iters):
    cdef uintptr_t p = _addr(mp), i = _addr(S_in), o = _addr(S_out), t = _addr(thresholds)
    cdef int rc
    with nogil:
        rc = c.lws_multi_batch_lws(<c.lws_multi_plan *>p, wsel, <const double *>i, <double *>o, B, T, <const double *>t, iters)
    return rc


def lws_multi_run_lws(mp, S_in, S_out, int B, int T, thr_nofuture, int it_nofuture, thr_online, int it_online, int LA,
                      qdiv_, thr_batch, int it_batch):
    cdef uintptr_t p = _addr(mp), i = _addr(S_in), o = _addr(S_out)
    cdef double qdiv = _dbl(qdiv_)
    cdef uintptr_t t0 = _addr(thr_nofuture), t1 = _addr(thr_online), t2 = _addr(thr_batch)
    cdef int rc
    with nogil:
        rc = c.lws_multi_run_lws(<c.lws_multi_plan *>p, <const double *>i, <double *>o, B, T, <const double *>t0, it_nofuture,
                                 <const double *>t1, it_online, LA, qdiv, <const double *>t2, it_batch)
    return rc


def lws_multi_residual(mp, S, int B, int T, out):
    cdef uintptr_t p = _addr(mp), s = _addr(S), o = _addr(out)
    cdef int rc
    with nogil:
        rc = c.lws_multi_residual(<c.lws_multi_plan *>p, <const double *>s, B, T, <double *>o)
    return rc
