/*
 * lws_hip.h -- C ABI of the MI355X-native LWS engine (liblws_hip.so).
 *
 * This is the drop-in boundary for the hot path of Jonathan-LeRoux/lws: everything the
 * reference's Python binding (python/lws.pyx) or its mex gateways (matlab/{batch,online,
 * nofuture}_lws.cpp) do between "I have a complex T x F spectrogram, weights and a threshold
 * schedule" and "here is the updated spectrogram".  Plain pointers and sizes only; no C++,
 * torch or HIP types appear in a signature (a stream is passed as void*).
 *
 * Each entry point names the reference interface it replaces.
 *
 * Conventions
 *   - spectrograms: B independent T x F complex matrices, row-major, bin fastest, frame next
 *     (numpy C order of a (B,T,F) array; identical memory to MATLAB's F x T column-major).
 *     F must be odd (non-negative frequencies of an even FFT) -- lws.pyx:223-224.
 *   - host entry points take/return complex128 (interleaved re,im doubles), exactly the dtype
 *     the reference returns (lws.pyx:212-213,256); device entry points work in place on
 *     complex64 (fp32 plans) or complex128 (fp64 plans) device memory.
 *   - weights: complex128 interleaved, C order [Qp][Q][L+1] as produced by create_weights
 *     (lws.pyx:160-181): Qp == Q ("summarised") or Qp == 2(F-1) ("general", the reference's
 *     fractionalQ kernels).  A weight participates iff |w| > 1e-12 (lws.pyx:231-232).
 *   - thresholds: `iters` doubles, NOT yet scaled; every spectrogram scales them by its own
 *     mean(|S|) as lws.pyx:240,245 / batch_lws.cpp:119-129 do.
 *   - every function returns LWS_OK (0) or an error code; lws_last_error() gives the text of the
 *     calling thread's most recent failure.  Nothing aborts, nothing throws.
 *   - a plan may be used from one thread at a time; distinct plans are independent.
 */
#ifndef LWS_HIP_H_
#define LWS_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lws_plan lws_plan; /* opaque */

enum {
    LWS_OK = 0,
    LWS_ERR_INVALID = 1,     /* bad shape / argument (the ValueErrors of lws.pyx) */
    LWS_ERR_HIP = 2,         /* a HIP runtime call failed */
    LWS_ERR_NOMEM = 3,
    LWS_ERR_UNSUPPORTED = 4  /* valid request this build cannot serve */
};

/* plan flags */
enum {
    LWS_PRECISION_FP32 = 0,        /* default: fp32 state, weights and accumulation */
    LWS_PRECISION_FP64 = 1,        /* fp64 everywhere (reference arithmetic): batch sweeps of Q = 2 / 4 plans (frames up to ~2090 bins) on the fp64 systolic engine (lws_sys64.hip), no-future and online sweeps of Q = 2 / 3 / 4 / 8 plans on LDS engines that are bit-identical to the order-exact generic engine (lws_nofuture.hip, lws_online64.hip), batch sweeps of the other plans with summarised tensors -- Q = 3, 5..8, 16, L up to 10 -- on the band engine (lws_band.hip, round 6), online sweeps of the other plans (and of Q = 8 plans) on the team engine's order-exact kernel (lws_team.hip: the generic engine's bits, ~10x faster), the rest on that engine */
    LWS_NOFUTURE_Q4_COMPAT = 2,    /* reproduce NoFuture_LWSQ4's addressing (lwslib.cpp:559-594) when Q == Qp == 4 */
    LWS_FORCE_GENERIC = 4,         /* never use the specialised systolic batch kernel */
    LWS_NO_DIRECT_IO = 8,          /* always go through the extended buffers (prep / extract passes), even for a call
                                      that is one batch stage on device complex64 data -- for comparison */
    LWS_GENERIC_PLAIN_LAYOUT = 32, /* generic engine: batch sweeps in the reference's layout instead of the time-skewed copy
                                      (same results bit for bit; for comparison) */
    LWS_STORAGE_FP16 = 16          /* fp16-complex storage (BASELINE config 5): between passes over HBM the batch kernel keeps
                                      the spectrogram as half2 and the target magnitudes as half (10 B instead of 20 B per
                                      active bin and sweep), scaled per spectrogram by the power of two that brings its
                                      largest magnitude to [1, 2); arithmetic stays fp32.  The reference has no such mode
                                      (every pointer of lwslib.h:6-26 is double*): tolerance in DESIGN.md section 6.
                                      Applies to batch sweeps the systolic kernel serves; I/O stays complex64 / complex128 */
};

/* which of the plan's weight tensors a call uses (class lws passes W_ai to nofuture_lws, lws.pyx:475) */
enum { LWS_W = 0, LWS_W_AI = 1, LWS_W_AF = 2 };

int lws_hip_version(void);                 /* 10000*major + 100*minor + patch */
const char *lws_last_error(void);
int lws_device_count(void);                /* GPUs visible to this process, 0 if none */

/* Replaces the per-call weight preparation of lws.pyx:227-232, 280-285, 341-352 and
 * batch_lws.cpp:78-105: uploads the three weight tensors once.  W_ai / W_af may be NULL if the
 * online path is not used.  `device` is a HIP device ordinal. */
int lws_plan_create(lws_plan **plan, int device, int F, int L, int Q, int Qp,
                    const double *W, const double *W_ai, const double *W_af, unsigned flags);
void lws_plan_destroy(lws_plan *plan);

/* lws.batch_lws(S, W, thresholds)            (lws.pyx:209-258; mex batch_lws.cpp:20-154)
 * S_out may alias S_in.  iters == 0 copies the input (lws.pyx:219-220). */
int lws_batch_lws(lws_plan *plan, int wsel, const double *S_in, double *S_out, int B, int T,
                  const double *thresholds, int iters);

/* lws.nofuture_lws(S, W, thresholds)         (lws.pyx:261-311; mex nofuture_lws.cpp:20-157) */
int lws_nofuture_lws(lws_plan *plan, int wsel, const double *S_in, double *S_out, int B, int T,
                     const double *thresholds, int iters);

/* lws.online_lws(S, W, W_ai, W_af, thresholds, LA, fshift)
 *                                            (lws.pyx:314-375; mex online_lws.cpp:21-181;
 *                                             driver TF_RTISI_LA lwslib.cpp:1424-1492)
 * qdiv is the reference's Qfloat = N / fshift (only read when update == 1, which no shipped caller
 * uses; kept for interface fidelity). */
int lws_online_lws(lws_plan *plan, const double *S_in, double *S_out, int B, int T,
                   const double *thresholds, int iters, int LA, double qdiv);

/* lws.lws.run_lws(S) = nofuture -> online -> batch (lws.pyx:495-499) without leaving the device.
 * Between stages the edge-pad frames are refreshed from the current first/last frame exactly as
 * three separate calls would (each call re-runs extspec, lws.pyx:155-156,235). A stage with
 * zero iterations is skipped.  Thresholds are per stage. */
int lws_run_lws(lws_plan *plan, const double *S_in, double *S_out, int B, int T,
                const double *thr_nofuture, int it_nofuture,
                const double *thr_online, int it_online, int LA, double qdiv,
                const double *thr_batch, int it_batch);

/* Device-resident variants: `S_dev` is a device pointer to B*T*F complex64 (fp32 plan) or
 * complex128 (fp64 plan) values, updated in place; `stream` is a hipStream_t (or NULL).
 * Work is enqueued on `stream`; the call returns without synchronising unless noted. */
int lws_batch_lws_dev(lws_plan *plan, int wsel, void *S_dev, int B, int T,
                      const double *thresholds, int iters, void *stream);
int lws_nofuture_lws_dev(lws_plan *plan, int wsel, void *S_dev, int B, int T,
                         const double *thresholds, int iters, void *stream);
int lws_online_lws_dev(lws_plan *plan, void *S_dev, int B, int T,
                       const double *thresholds, int iters, int LA, double qdiv, void *stream);

/* lws.lws.run_lws on device-resident spectrograms (in place), the three stages enqueued back to back on `stream`. */
int lws_run_lws_dev(lws_plan *plan, void *S_dev, int B, int T,
                    const double *thr_nofuture, int it_nofuture,
                    const double *thr_online, int it_online, int LA, double qdiv,
                    const double *thr_batch, int it_batch, void *stream);

/* Host-only query (no device needed): does the weight tensor W[Qp][Q][L+1] (complex128 interleaved, as create_weights returns it,
 * lws.pyx:160-181) have the structure the fast kernels rely on -- W[p][r][k] == W[0][r][k] exp(2 pi j p r step / period) for every row
 * p?  Returns 1 and sets *period, *step (period 0: the tensor has no neighbour-frame weights and fits any twiddle), else 0.  For
 * create_weights' tensors period / step = frame / hop in lowest terms; plans whose tensors have it run on the systolic / LDS engines
 * when their shape is served, the others on the generic engine. */
int lws_weights_structure(const double *W, int Q, int Qp, int L, int *period, int *step);

/* Pre-size every scratch buffer of the plan for calls of up to B spectrograms of T frames and `max_iters` thresholds
 * per stage.  The *_dev entry points only enqueue work and return -- unless a scratch buffer has to grow, which is a
 * (synchronising) hipMalloc: reserve once and they never allocate.  The reference allocates per call (lws.pyx:227-240;
 * the mex gateways malloc and never free, batch_lws.cpp:81-118). */
int lws_plan_reserve(lws_plan *plan, int B, int T, int max_iters);

/* Consistency-residual proxy: for each spectrogram b
 *   out[2b]   = sum over bins of |acc + w00*S|^2   (acc = the LWS weighted sum, w00 = W[0][0][0])
 *   out[2b+1] = sum over bins of |S|^2
 * in fp64, on the device buffer `S_dev`.  out is a HOST array of 2*B doubles (synchronises). */
int lws_residual_dev(lws_plan *plan, const void *S_dev, int B, int T, double *out, void *stream);
/* The same for HOST spectrograms S[B][T][F] complex128. */
int lws_residual(lws_plan *plan, const double *S, int B, int T, double *out);

/* The JOB's residual pair for a caller that runs ONE PROCESS PER GPU (the layout bench.py uses; BASELINE.json north_star: "RCCL
 * over xGMI used only for the optional final consistency-residual reduction"): this rank's spectrograms are summed on the device
 *   local[0] = sum_b sum over bins |acc + w00*S|^2,   local[1] = sum_b sum over bins |S|^2      (fp64, fixed order)
 * and the pair is all-reduced (ncclSum over 2 doubles, in place on the device, on `stream`) over the ranks of `rccl_comm` -- an
 * ncclComm_t the caller created (ncclCommInitRank) for its ranks; NULL: no collective, out = this rank's sums.  out: 2 HOST doubles,
 * the same on every rank (synchronises).  RCCL is loaded at the first call (dlopen librccl.so.1): the library does not link
 * against it, and callers that never pass a communicator never need it (LWS_ERR_UNSUPPORTED if it cannot be loaded).  The reference
 * has no counterpart (single process, no residual); the single-process multi-GPU form is lws_multi_residual below. */
int lws_residual_allreduce_dev(lws_plan *plan, const void *S_dev, int B, int T, void *rccl_comm, double *out, void *stream);

/* Timing of the most recent *_dev / host call on this plan, measured with HIP events on the
 * stream the kernels ran on: total milliseconds spent in the update kernels and the number of
 * update-kernel launches (prep / extract kernels are not counted). */
int lws_last_kernel_time(lws_plan *plan, float *ms, int *launches);

/* Device-to-device stream copy of `bytes` (a multiple of 16) with the library's own kernel: the measured HBM copy
 * rate the roofline fraction is quoted beside (spec peak and measured copy peak). */
int lws_stream_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream);

/* Name of the update kernel the last call dispatched ("generic_fp32", "systolic_q4", ...). */
const char *lws_last_kernel_name(lws_plan *plan);

/* Which stage of this plan's calls ("batch", "no-future", "online") last ran on the order-exact generic engine -- 20-40x slower than
 * the kernels built for the common shapes -- or "" if none ever has.  (lws_last_kernel_name only names the LAST stage of a
 * run_lws pipeline.) */
const char *lws_generic_stage(lws_plan *plan);

/* ---- the steps either side of the path, on the device (lws.pyx:43-144; float32, any even frame size N in [32, 4096]
 *      -- an odd factor times a power of two: radix-2 stages and one stage of odd-point DFTs; fftsize == fsize unless said otherwise).  Windows are host arrays of N doubles, already normalised the way the caller
 *      wants them (class lws: awin and synthwin(awin, fshift)).  perfectrec as in lws.pyx:55-67,130-137. ---- */

/* Frames stft() produces for a signal of `len` samples (lws.pyx:55-76); < 1 if the signal is too short. */
int lws_stft_frames(int len, int N, int fshift, int perfectrec);
/* Samples istft() returns for M frames (lws.pyx:121,130-137). */
int lws_istft_length(int M, int N, int fshift, int perfectrec);
/* stft (lws.pyx:43-90) of B signals x_dev[B][len] (float32) into S_dev[B][M][N/2+1] (complex64), M = lws_stft_frames(). */
int lws_stft_dev(int device, const float *x_dev, int B, int len, int N, int fshift, const double *awin,
                 int perfectrec, void *S_dev, void *stream);
/* istft (lws.pyx:93-137) of S_dev[B][M][N/2+1] (complex64) into x_dev[B][lws_istft_length()] (float32). */
int lws_istft_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *swin,
                  int perfectrec, float *x_dev, void *stream);
/* stft with a transform longer than the frame (lws.pyx:49-50,85: np.fft.fft(frame, n = fftsize) of the fsize windowed samples,
 * zeros behind them): fsize even, fsize <= fftsize, fftsize an even size in [32, 4096]; frame count and padding follow fsize
 * (M = lws_stft_frames(len, fsize, ...)); awin has fsize entries; S_dev[B][M][fftsize/2+1].  (There is no inverse counterpart: the
 * reference's istft raises for every fftsize != 2 (bins - 1), lws.pyx:107-126; class lws pads its windows instead, 396-411.) */
int lws_stft_zp_dev(int device, const float *x_dev, int B, int len, int fsize, int fftsize, int fshift, const double *awin,
                    int perfectrec, void *S_dev, void *stream);
/* get_consistency (lws.pyx:140-144) per spectrogram: out[2b] = sum |S|^2, out[2b+1] = sum |stft(istft(S)) - S|^2
 * (fp64 sums of the fp32 transform); consistency in dB = 10 log10(out[2b] / out[2b+1]).  out: HOST, 2*B doubles
 * (synchronises).  Sums of several spectrograms / ranks add up to the batch consistency. */
int lws_consistency_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *awin,
                        const double *swin, int perfectrec, double *out, void *stream);

/* ---- host-side construction of windows, weights and schedules (lws.pyx:10-40,160-206), fp64, no device work: what a
 *      caller without numpy needs to build a plan.  Complex outputs are interleaved (re, im) doubles. ---- */

/* hann (lws.pyx:10-19): out[n]. */
int lws_hann(int n, int symmetric, int use_offset, double *out);
/* synthwin (lws.pyx:22-40): synthesis window normalised for perfect reconstruction; swin may be NULL (= awin). */
int lws_synthwin(const double *awin, int fsize, int fshift, const double *swin, double *out);
/* shape of create_weights' result: Q = ceil(fsize/fshift), Qprime = Q (summarised, shift divides the window) or fsize. */
int lws_weights_shape(int fsize, int fshift, int use_summarized_weights, int *Qprime, int *Q);
/* create_weights (lws.pyx:160-181): W[Qprime][Q][L+1] complex128. */
int lws_create_weights(const double *awin, const double *swin, int fsize, int fshift, int L, int use_summarized_weights,
                       double *W);
/* build_asymmetric_windows (lws.pyx:184-200) from the product awin*swin: win_ai[fsize], win_af[fsize]. */
int lws_build_asymmetric_windows(const double *awin_swin, int fsize, int fshift, double *win_ai, double *win_af);
/* get_thresholds (lws.pyx:203-206): out[i] = alpha * exp(-beta * i^gamma). */
int lws_get_thresholds(int iterations, double alpha, double beta, double gamma, double *out);
/* What `lws.lws(awin_or_fsize, fshift, L=..., swin=...)` sets up (lws.pyx:384-431): awin == NULL means the default
 * sqrt-Hann window of `fsize` samples; swin may be NULL; the three weight tensors W, W_ai, W_af are built and the plan
 * created.  awin_out / swin_out (may be NULL) receive the windows actually used, fsize doubles each. */
int lws_plan_create_from_windows(lws_plan **plan, int device, const double *awin, const double *swin, int fsize,
                                 int fshift, int L, int symmetric_win, unsigned flags, double *awin_out, double *swin_out);

/* ---- one node, several GPUs, no torch: independent spectrograms are dealt in contiguous blocks to one plan per device,
 *      each driven by its own host thread and stream; no exchange between devices during the sweeps.
 *      `devices`: ndev HIP ordinals, or NULL for devices 0..ndev-1; ndev <= 0 means every visible device.  A device may
 *      be listed more than once (several shards on one GPU).  What a mex gateway or a C++ caller uses where the Python
 *      layer uses one process per GPU (bench.py). ---- */
typedef struct lws_multi_plan lws_multi_plan; /* opaque */
int lws_multi_plan_create(lws_multi_plan **mp, int ndev, const int *devices, int F, int L, int Q, int Qp,
                          const double *W, const double *W_ai, const double *W_af, unsigned flags);
void lws_multi_plan_destroy(lws_multi_plan *mp);
int lws_multi_plan_shards(const lws_multi_plan *mp);
/* lws_batch_lws / lws_run_lws over all shards; S_out may alias S_in; same results as one plan on one device. */
int lws_multi_batch_lws(lws_multi_plan *mp, int wsel, const double *S_in, double *S_out, int B, int T,
                        const double *thresholds, int iters);
int lws_multi_run_lws(lws_multi_plan *mp, const double *S_in, double *S_out, int B, int T,
                      const double *thr_nofuture, int it_nofuture,
                      const double *thr_online, int it_online, int LA, double qdiv,
                      const double *thr_batch, int it_batch);
/* The job's consistency-residual pair (see lws_residual_dev) of HOST spectrograms S[B][T][F] complex128, summed over all
 * shards on the host: out[0] = sum |acc + w00 S|^2, out[1] = sum |S|^2. */
int lws_multi_residual(lws_multi_plan *mp, const double *S, int B, int T, double *out);

#ifdef __cplusplus
}
#endif
#endif
