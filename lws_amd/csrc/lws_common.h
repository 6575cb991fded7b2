// lws_common.h -- internal declarations shared by the HIP translation units of liblws_hip.so.
// Not part of the public ABI (that is include/lws_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace lws {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device; the launchers keep one bit per device in a function-local
// atomic (lws_multi_* launches the same kernels from one host thread per device).  Returns true if this device still needs
// the call; a lost race only repeats it.
inline bool attr_needed(std::atomic<unsigned long long> &done, int *dev_out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *dev_out = dev;
    return dev < 0 || dev >= 64 || !((done.load(std::memory_order_relaxed) >> dev) & 1ull);
}
inline void attr_done(std::atomic<unsigned long long> &done, int dev) {
    if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
}

// records the text lws_last_error() returns and passes `code` through (lws_capi.hip)
int set_error(int code, const char *fmt, ...);

// CPUs this process can actually use at once: the CPUs it may run on, capped by the container's CPU quota (cgroup cpu.max -- the GPU
// boxes of this pool give a container 16 CPUs of their 256 hardware threads; std::thread::hardware_concurrency() does not see that).
int usable_cpus();

template <typename real> struct cx;
template <> struct cx<float>  { using type = float2; };
template <> struct cx<double> { using type = double2; };

// What a sweep does, in the reference's terms.
//   BATCH    : LWSQ2/LWSQ4/LWSanyQ/LWSfractionalQ            (lwslib.cpp:72-467)
//   NOFUTURE : NoFuture_LWS{Q2,anyQ,fractionalQ}              (lwslib.cpp:473-535,620-764)
//   NOFUTURE_Q4_COMPAT : NoFuture_LWSQ4 with its flat-offset addressing (lwslib.cpp:538-617)
//   ONLINE   : TF_RTISI_LA and the Asym_UpdatePhase* calls it makes (lwslib.cpp:776-1492)
enum Mode { MODE_BATCH = 0, MODE_NOFUTURE = 1, MODE_NOFUTURE_Q4_COMPAT = 2, MODE_ONLINE = 3,
            MODE_ASYM = 4 /* one Asym_UpdatePhase* call: T frames that may read M0 frames to their right */ };

// Device weights of one weight tensor: [Qp][Q][L+1], entries with |w| <= 1e-12 have flag 0.
template <typename real> struct WeightSet {
    const typename cx<real>::type *w;
    const uint8_t *flag;
};

// Arguments of the order-exact generic wavefront kernel (lws_generic.hip).
template <typename real> struct GenericArgs {
    typename cx<real>::type *state;  // [B][Tp][Np] extended spectrograms (the reference's ExtSr/ExtSi)
    const real *amp;                 // [B][Tp][Np] target magnitudes (AmpSpec)
    const real *thr;                 // [B][n_thr]  thresholds already scaled by mean|S| of each spectrogram
    WeightSet<real> w[3];            // LWS_W, LWS_W_AI, LWS_W_AF
    int wsel;                        // weight tensor used by batch / no-future sweeps
    int F, T, L, Q, Qp;
    int n_thr;                       // sweeps (batch / no-future) or iterations per frame (online)
    int LA;                          // look-ahead (online)
    int M0;                          // frames usable to the right of the first one (MODE_ASYM)
    int update;                      // 1: add S/qdiv to the centre sum (dead in shipped callers), 2: do not
    real qdiv;
    int mode;
    int group;                       // max sweeps in flight (batch / no-future)
    int mask_rows;                   // set by launch_generic: weight rows whose flags are packed into LDS masks (0: none)
};

template <typename real>
hipError_t launch_generic(const GenericArgs<real> &a, int B, hipStream_t stream);
// Batch sweeps (mode == MODE_BATCH) on a time-skewed copy of the state in the scratch buffers sw / aw (sizes from
// generic_skew_bytes): same results as launch_generic, bit for bit, with coalesced tap loads.
template <typename real>
hipError_t launch_generic_skewed(const GenericArgs<real> &a, int B, void *sw, void *aw, hipStream_t stream);
template <typename real>
size_t generic_skew_bytes(int B, int F, int T, int L, int Q, size_t *amp_bytes);

// ---- prep / extract (lws_generic.hip) ----
// in: [B][T][F] complex (double2 or float2).  Builds the extended buffer, |.|, per-spectrogram
// mean|S| and the scaled threshold table.  row_sums: scratch [B][T] doubles.
template <typename real, typename in_cx>
hipError_t launch_prep(const in_cx *in, typename cx<real>::type *state, real *amp,
                       double *row_sums, double *mean_amp, int B, int T, int F, int L, int Q,
                       hipStream_t stream);
template <typename real>
hipError_t launch_scale_thresholds(const double *thr, const double *mean_amp, real *out, int B,
                                   int n, hipStream_t stream);
// Re-creates the edge-pad frames from the current first / last frame (what a fresh extspec()
// call would do, lws.pyx:155-156) and recomputes |.| and mean|S| from the current state.
template <typename real>
hipError_t launch_refresh(typename cx<real>::type *state, real *amp, double *row_sums,
                          double *mean_amp, int B, int T, int F, int L, int Q, hipStream_t stream);
// out: [B][T][F]; if `orig` is non-null, bins whose state still equals the rounded original are
// returned as the original value (never-updated bins stay bit-identical).
template <typename real, typename out_cx>
hipError_t launch_extract(const typename cx<real>::type *state, out_cx *out, const out_cx *orig,
                          int B, int T, int F, int L, int Q, hipStream_t stream);

// rows: scratch [B][T][2] doubles; out: device [B][2] doubles.
template <typename real>
hipError_t launch_residual(const typename cx<real>::type *state, WeightSet<real> ws, double *rows,
                           double *out, int B, int T, int F, int L, int Q, int Qp,
                           hipStream_t stream);

// Twiddle structure of a weight tensor W[Qp][Q][L+1] (host, complex128 interleaved): W[p][r][k] == W[0][r][k] exp(2 pi j p r s / P)
// for every row p, with the smallest such P <= pmax (lws_online.hip).  create_weights (lws.pyx:160-181) builds exactly that:
// s / P = hop / frame in lowest terms -- P = Q, s = 1 when the hop divides the frame, for summarised (Qp = Q) and general (Qp = N) tensors.
bool weights_twiddle(const double *W, int Q, int Qp, int L, int pmax, int *P_out, int *s_out);

}  // namespace lws
