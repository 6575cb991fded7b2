// lws_systolic.h -- interface of the systolic batch-LWS kernel (lws_systolic.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace lws {

constexpr int SYSTOLIC_MAX_ITERS = 440;  // thresholds of one launch live in LDS; longer schedules run as several launches

struct SystolicPlan {
    bool ok[3] = {false, false, false};  // per weight tensor: kernel applicable
    int F = 0, L = 0, Q = 0;
    int Lk = 0;               // stencil half-width of the kernel build that serves the plan: L, or L + 1 for an even L (the
                              // windows are fetched as pairs of bins; the extra tap has weight zero and is never fetched)
    void *tables[3] = {nullptr, nullptr, nullptr};  // device weight tables
    void *sk_state = nullptr, *sk_amp = nullptr;    // skewed-layout scratch
    size_t sk_state_cap = 0, sk_amp_cap = 0;
    const char *name = "systolic";
    char name_buf[64] = {0};
    int *err_dev = nullptr;   // device flag of the last launch: a workgroup gave up waiting for its producer (the call was
                              // then re-run with one workgroup per spectrogram, on the device, before it completed)
    int last_nwg = 1;         // workgroups per spectrogram of the last launch
    bool h16 = false;         // fp16-complex storage of the skewed layout (LWS_STORAGE_FP16)
    void *thr_chunk = nullptr;   // dense threshold table of one launch of a schedule longer than SYSTOLIC_MAX_ITERS
    size_t thr_chunk_cap = 0;
};

// Analyse the (host, complex128 interleaved) weight tensors and upload tables for the ones the
// kernel can serve.  Never fails for "not applicable"; only for HIP errors.
hipError_t systolic_build(SystolicPlan &sp, int F, int L, int Q, int Qp, const double *const W[3], bool fp16_storage);
void systolic_release(SystolicPlan &sp);
bool systolic_supports(const SystolicPlan &sp, int wsel, int T);
// Allocates the skewed-layout scratch for calls of up to B spectrograms x T frames (so that later calls do not).
hipError_t systolic_reserve(SystolicPlan &sp, int B, int T, int iters);
const char *systolic_name(const SystolicPlan &sp);
// Runs `iters` batch sweeps on the extended buffers (reference layout), in place.
// ev0/ev1 (may be null) are recorded around the update kernel(s) only.
hipError_t launch_systolic(SystolicPlan &sp, int wsel, float2 *state, const float *amp,
                           const float *thr, int B, int T, int iters, hipStream_t stream,
                           int *launches, hipEvent_t ev0, hipEvent_t ev1);
// A call that consists of one batch stage on device complex64 spectrograms [B][T][F] skips the extended buffers:
// systolic_io_load converts `in` straight to the kernel's layout and computes mean|S| (partial: scratch of
// B * systolic_io_partials() doubles), the caller scales the thresholds, systolic_io_run runs the sweeps and writes `out`.
size_t systolic_io_partials(const SystolicPlan &sp, int T);
hipError_t systolic_io_load(SystolicPlan &sp, const float2 *in, int B, int T, int iters, double *partial, double *mean_amp,
                            hipStream_t stream);
// (`in` and `partial` as given to systolic_io_load: a failed multi-workgroup hand-over re-converts from them; `in` may be `out`)
hipError_t systolic_io_run(SystolicPlan &sp, int wsel, const float *thr, const float2 *in, float2 *out, double *partial, int B,
                           int T, int iters, hipStream_t stream, int *launches, hipEvent_t ev0, hipEvent_t ev1);

// One compilation of lws_systolic.hip, as a table: a plan is served by the first build whose systolic_build() accepts its
// shape and weights (lws_capi.hip tries them in the order below).
struct SystolicBuild {
    decltype(&systolic_build) build;
    decltype(&systolic_release) release;
    decltype(&systolic_supports) supports;
    decltype(&systolic_reserve) reserve;
    decltype(&systolic_name) name;
    decltype(&launch_systolic) launch;
    decltype(&systolic_io_partials) io_partials;
    decltype(&systolic_io_load) io_load;
    decltype(&systolic_io_run) io_run;
};
const SystolicBuild &systolic_entry();                            // Q in {2, 4}, frames of up to 513 bins: 7 sweep slots
namespace q8 { const SystolicBuild &systolic_entry(); }           // -DLWS_Q8=1: Q = 8, 64-step ring, 2 sweep slots
namespace wide { const SystolicBuild &systolic_entry(); }         // -DLWS_WIDE=1: frames of up to 1025 bins, two waves per sweep
                                                                  // slot, 3 slots
namespace xwide { const SystolicBuild &systolic_entry(); }        // -DLWS_WIDE=2: frames of up to 2049 bins, four waves per sweep slot, 1 slot
namespace q2 { const SystolicBuild &systolic_entry(); }           // -DLWS_R16=1: Q = 2 with a 16-step ring: 15 sweep slots, frames <= 513 bins
namespace wide_q2 { const SystolicBuild &systolic_entry(); }      // -DLWS_WIDE=1 -DLWS_R16=1: the same for frames of up to 1025 bins (7 slots of two waves)
namespace half_q2 { const SystolicBuild &systolic_entry(); }      // -DLWS_SPW=2 -DLWS_R16=1: ... of up to 257 bins (26 slots on 13 waves)
namespace quarter_q2 { const SystolicBuild &systolic_entry(); }   // -DLWS_SPW=4 -DLWS_R16=1: ... of up to 129 bins (44 slots on 11 waves)
namespace l7 { const SystolicBuild &systolic_entry(); }           // -DLWS_L7=1: L = 6, 7 (frames 16 steps apart, 64-step ring, 3 slots), <= 513 bins
namespace half { const SystolicBuild &systolic_entry(); }         // -DLWS_SPW=2: frames of up to 257 bins, two sweep slots per wave (14)
namespace quarter { const SystolicBuild &systolic_entry(); }      // -DLWS_SPW=4: frames of up to 129 bins, four sweep slots per wave (24)
namespace tw { const SystolicBuild &systolic_entry(); }           // -DLWS_TW=1: twiddles from a table -- Q = 3, general weights of a fractional Q (Q <= 4), <= 513 bins
namespace tw_wide { const SystolicBuild &systolic_entry(); }      // -DLWS_TW=1 -DLWS_WIDE=1: the same for frames of up to 1025 bins (two waves per sweep slot)
namespace tw_q8 { const SystolicBuild &systolic_entry(); }        // -DLWS_TW=1 -DLWS_Q8=1: table twiddles on the 64-step ring with helper waves: ceil(frame/hop) in 5..8
namespace tw_q5 { const SystolicBuild &systolic_entry(); }        // ... -DLWS_TWQ=5: exactly 5 frames per stencil row on a 40-step ring: 3 sweep slots of a main and a helper wave
namespace tw_q6 { const SystolicBuild &systolic_entry(); }        // ... -DLWS_TWQ=6: 6 frames per row, 48-step ring, 3 slots
namespace tw_half { const SystolicBuild &systolic_entry(); }      // -DLWS_TW=1 -DLWS_SPW=2: the same for frames of up to 257 bins (25 ms / 10 ms speech framing)

}  // namespace lws
