# lws_hip.pxd -- Cython declaration of the C ABI (include/lws_hip.h): what python/lwslib.pxd:1-13 is to lwslib.h.
# Every entry point is nogil: the reference holds the GIL for a whole call (its generated C++ has no
# Py_BEGIN_ALLOW_THREADS); here a call only enqueues device work or waits for it.
from libc.stddef cimport size_t

cdef extern from "lws_hip.h" nogil:
    ctypedef struct lws_plan:
        pass
    ctypedef struct lws_multi_plan:
        pass
    int lws_hip_version()
    const char *lws_last_error()
    int lws_device_count()
    int lws_plan_create(lws_plan **plan, int device, int F, int L, int Q, int Qp, const double *W, const double *W_ai,
                        const double *W_af, unsigned flags)
    void lws_plan_destroy(lws_plan *plan)
    int lws_batch_lws(lws_plan *plan, int wsel, const double *S_in, double *S_out, int B, int T, const double *thresholds,
                      int iters)
    int lws_nofuture_lws(lws_plan *plan, int wsel, const double *S_in, double *S_out, int B, int T,
                         const double *thresholds, int iters)
    int lws_online_lws(lws_plan *plan, const double *S_in, double *S_out, int B, int T, const double *thresholds, int iters,
                       int LA, double qdiv)
    int lws_run_lws(lws_plan *plan, const double *S_in, double *S_out, int B, int T, const double *thr_nofuture,
                    int it_nofuture, const double *thr_online, int it_online, int LA, double qdiv, const double *thr_batch,
                    int it_batch)
    int lws_batch_lws_dev(lws_plan *plan, int wsel, void *S_dev, int B, int T, const double *thresholds, int iters,
                          void *stream)
    int lws_nofuture_lws_dev(lws_plan *plan, int wsel, void *S_dev, int B, int T, const double *thresholds, int iters,
                             void *stream)
    int lws_online_lws_dev(lws_plan *plan, void *S_dev, int B, int T, const double *thresholds, int iters, int LA,
                           double qdiv, void *stream)
    int lws_run_lws_dev(lws_plan *plan, void *S_dev, int B, int T, const double *thr_nofuture, int it_nofuture,
                        const double *thr_online, int it_online, int LA, double qdiv, const double *thr_batch, int it_batch,
                        void *stream)
    int lws_plan_reserve(lws_plan *plan, int B, int T, int max_iters)
    int lws_residual_dev(lws_plan *plan, const void *S_dev, int B, int T, double *out, void *stream)
    int lws_residual(lws_plan *plan, const double *S, int B, int T, double *out)
    int lws_weights_structure(const double *W, int Q, int Qp, int L, int *period, int *step)
    int lws_residual_allreduce_dev(lws_plan *plan, const void *S_dev, int B, int T, void *rccl_comm, double *out, void *stream)
    int lws_last_kernel_time(lws_plan *plan, float *ms, int *launches)
    const char *lws_last_kernel_name(lws_plan *plan)
    const char *lws_generic_stage(lws_plan *plan)
    int lws_stream_copy(void *dst_dev, const void *src_dev, size_t nbytes, void *stream)
    int lws_stft_frames(int length, int N, int fshift, int perfectrec)
    int lws_istft_length(int M, int N, int fshift, int perfectrec)
    int lws_stft_dev(int device, const float *x_dev, int B, int length, int N, int fshift, const double *awin,
                     int perfectrec, void *S_dev, void *stream)
    int lws_istft_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *swin, int perfectrec,
                      float *x_dev, void *stream)
    int lws_stft_zp_dev(int device, const float *x_dev, int B, int length, int fsize, int fftsize, int fshift, const double *awin,
                        int perfectrec, void *S_dev, void *stream)
    int lws_consistency_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *awin,
                            const double *swin, int perfectrec, double *out, void *stream)
    int lws_hann(int n, int symmetric, int use_offset, double *out)
    int lws_synthwin(const double *awin, int fsize, int fshift, const double *swin, double *out)
    int lws_weights_shape(int fsize, int fshift, int use_summarized_weights, int *Qprime, int *Q)
    int lws_create_weights(const double *awin, const double *swin, int fsize, int fshift, int L,
                           int use_summarized_weights, double *W)
    int lws_build_asymmetric_windows(const double *awin_swin, int fsize, int fshift, double *win_ai, double *win_af)
    int lws_get_thresholds(int iterations, double alpha, double beta, double gamma, double *out)
    int lws_plan_create_from_windows(lws_plan **plan, int device, const double *awin, const double *swin, int fsize,
                                     int fshift, int L, int symmetric_win, unsigned flags, double *awin_out,
                                     double *swin_out)
    int lws_multi_plan_create(lws_multi_plan **mp, int ndev, const int *devices, int F, int L, int Q, int Qp,
                              const double *W, const double *W_ai, const double *W_af, unsigned flags)
    void lws_multi_plan_destroy(lws_multi_plan *mp)
    int lws_multi_plan_shards(const lws_multi_plan *mp)
    int lws_multi_batch_lws(lws_multi_plan *mp, int wsel, const double *S_in, double *S_out, int B, int T,
                            const double *thresholds, int iters)
    int lws_multi_run_lws(lws_multi_plan *mp, const double *S_in, double *S_out, int B, int T, const double *thr_nofuture,
                          int it_nofuture, const double *thr_online, int it_online, int LA, double qdiv,
                          const double *thr_batch, int it_batch)
    int lws_multi_residual(lws_multi_plan *mp, const double *S, int B, int T, double *out)
