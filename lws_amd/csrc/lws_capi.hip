// lws_capi.hip -- the extern "C" boundary declared in include/lws_hip.h: plan management, buffer
// ownership, dispatch between the generic order-exact kernel and the systolic batch kernel, and
// HIP-event timing of the update kernels.
#include "../../include/lws_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "lws_common.h"
#include <type_traits>
#include "lws_systolic.h"
#include "lws_sys64.h"
#include <sched.h>
#include <cstring>
#include "lws_band.h"
#include "lws_online.h"
#include "lws_online64.h"
#include "lws_team.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <dlfcn.h>
#ifdef MADV_POPULATE_WRITE
#define LWS_MADV_POPULATE_WRITE MADV_POPULATE_WRITE
#else
#define LWS_MADV_POPULATE_WRITE 23   // Linux 5.14; older kernels answer EINVAL and the pages are faulted in by their first use
#endif
#include "lws_nofuture.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(LWS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

}  // namespace

namespace lws {
// error text for lws_last_error(), shared with the other translation units of the library
int set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace lws

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return LWS_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail(LWS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return LWS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

// Staging of the complex128 host entry points of an fp32 plan (run_host_pipelined): pinned host buffers for complex64
// chunks on their way up and down, the device buffers the chunks are processed in (in place), copy / compute streams.
struct HostPipe {
    void *up[2] = {nullptr, nullptr}, *down[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    DevBuf io[2];
    DevBuf io_real[2];                            // a chunk whose input is real-valued goes up as 4 B per bin and is expanded here
    hipStream_t s_up = nullptr, s_down = nullptr, s_comp = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
    hipEvent_t ev_load[2] = {nullptr, nullptr};   // a chunk's light first kernels (layout, mean, thresholds) are done
    static constexpr int PIECES = 4;              // the first upload and the last download go in pieces, the host pass of a piece
    hipEvent_t ev_piece[PIECES] = {nullptr, nullptr, nullptr, nullptr};   // ... beside the copy of its neighbour
    int ensure(size_t bytes, int slots = 2) {   // slots: 1 for a call that is a single chunk
        if (!s_up) {
            if (hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking) != hipSuccess)
                return fail(LWS_ERR_HIP, "hipStreamCreate failed");
            for (int i = 0; i < 2; ++i)
                if (hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ev_down[i], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&ev_load[i], hipEventDisableTiming) != hipSuccess)
                    return fail(LWS_ERR_HIP, "hipEventCreate failed");
            for (int i = 0; i < PIECES; ++i)
                if (hipEventCreateWithFlags(&ev_piece[i], hipEventDisableTiming) != hipSuccess) return fail(LWS_ERR_HIP, "hipEventCreate failed");
        }
        for (int i = 0; i < slots && i < 2; ++i) {
            if (bytes > cap[i]) {
                if (up[i]) (void)hipHostFree(up[i]);
                if (down[i]) (void)hipHostFree(down[i]);
                up[i] = down[i] = nullptr;
                cap[i] = 0;
                if (hipHostMalloc(&up[i], bytes, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&down[i], bytes, hipHostMallocDefault) != hipSuccess)
                    return fail(LWS_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
                cap[i] = bytes;
            }
            int rc = io[i].ensure(bytes);
            if (rc) return rc;
            if ((rc = io_real[i].ensure(bytes / 2))) return rc;
        }
        return LWS_OK;
    }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (up[i]) (void)hipHostFree(up[i]);
            if (down[i]) (void)hipHostFree(down[i]);
            up[i] = down[i] = nullptr;
            io[i].release();
            io_real[i].release();
            if (ev_up[i]) (void)hipEventDestroy(ev_up[i]);
            if (ev_comp[i]) (void)hipEventDestroy(ev_comp[i]);
            if (ev_down[i]) (void)hipEventDestroy(ev_down[i]);
            if (ev_load[i]) (void)hipEventDestroy(ev_load[i]);
            ev_up[i] = ev_comp[i] = ev_down[i] = ev_load[i] = nullptr;
        }
        cap[0] = cap[1] = 0;
        for (int i = 0; i < PIECES; ++i) {
            if (ev_piece[i]) (void)hipEventDestroy(ev_piece[i]);
            ev_piece[i] = nullptr;
        }
        if (s_up) (void)hipStreamDestroy(s_up);
        if (s_down) (void)hipStreamDestroy(s_down);
        if (s_comp) (void)hipStreamDestroy(s_comp);
        s_up = s_down = s_comp = nullptr;
    }
};

struct lws_plan {
    int device = 0;
    int F = 0, L = 0, Q = 0, Qp = 0;
    unsigned flags = 0;
    bool fp64 = false;
    bool have[3] = {false, false, false};
    int wperiod[3] = {0, 0, 0};    // period of the weight rows of each tensor (Q for a summarised one; 0: rows do not repeat)
    bool twiddle_all = false;      // W, W_ai and W_af all have create_weights' twiddle structure (the online LDS engine relies on it)
    int tw_P = 0, tw_s = 0;        // ... W[p][r][k] = W[0][r][k] exp(2 pi j p r tw_s / tw_P), the same for the three of them
    DevBuf online_tw;              // the online engine's twiddle table when those are not the static eighth turns of Q in {2,4,8}
    std::vector<double> hostW[3];  // complex128 interleaved copies (eligibility analysis, systolic tables)
    DevBuf w[3], wflag[3];
    DevBuf state, amp, row_sums, mean_amp, thr_host_copy, thr_scaled, stage, resid_rows, resid_out, resid_sum;
    DevBuf gsk_state, gsk_amp;     // time-skewed copy of the state for the generic engine's batch sweeps
    DevBuf band_tab[3];            // the band engine's tables of each weight tensor (lws_band.h: band_tables), uploaded at first use
    int band_tab_lt[3] = {0, 0, 0};
    HostPipe pipe;                 // host-array entry points: pinned staging, chunk buffers, streams
    lws::SystolicPlan sys;         // device tables of the systolic kernel (empty if not eligible)
    const lws::SystolicBuild *sysb = nullptr;   // the build of it that serves this plan (narrow / Q = 8 / wide), if any
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *host_pool = nullptr;     // HostWorkers of the host-array entry points (kept between calls: 32 thread starts cost ~1 ms)
    int host_pool_n = 0;
    int host_threads = 0;          // > 0: the conversion threads this plan may use (lws_multi_*: an even share of the usable CPUs per device)
    hipEvent_t ev_after_load = nullptr;   // if set: recorded by run_pipeline between its light first kernels and the update kernels
    // The plan's scratch (state, amp, thresholds, the skewed layouts, progress counters) is shared by all of its calls.  A *_dev
    // call only enqueues work; the next call of the plan -- on another stream, or a host-array call on the pipeline's private
    // streams -- must run after it: every call records ev_busy behind its last kernel and waits for the previous one's.
    hipEvent_t ev_busy = nullptr;
    bool busy = false;
    bool timing_pending = false;
    float last_ms = 0.f;
    int last_launches = 0;
    const char *last_name = "none";
    const char *generic_stage = "";   // the stage ("batch", "no-future", "online") that last ran on the generic engine; "" if none has
};

namespace {

template <typename real>
int upload_weights(lws_plan *p, int which, const double *W) {
    using C = typename lws::cx<real>::type;
    const size_t n = (size_t)p->Qp * p->Q * (p->L + 1);
    std::vector<C> w(n);
    std::vector<uint8_t> f(n);
    for (size_t i = 0; i < n; ++i) {
        const double re = W[2 * i], im = W[2 * i + 1];
        const bool on = std::hypot(re, im) > 1.0e-12;  // lws.pyx:231-232
        f[i] = on ? 1 : 0;
        w[i].x = on ? (real)re : (real)0;
        w[i].y = on ? (real)im : (real)0;
    }
    int rc = p->w[which].ensure(n * sizeof(C));
    if (rc) return rc;
    rc = p->wflag[which].ensure(n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p->w[which].p, w.data(), n * sizeof(C), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->wflag[which].p, f.data(), n, hipMemcpyHostToDevice));
    p->hostW[which].assign(W, W + 2 * n);
    p->have[which] = true;
    return LWS_OK;
}

template <typename real>
lws::WeightSet<real> wset(const lws_plan *p, int which) {
    lws::WeightSet<real> s;
    s.w = static_cast<const typename lws::cx<real>::type *>(p->w[which].p);
    s.flag = static_cast<const uint8_t *>(p->wflag[which].p);
    return s;
}

int check_common(const lws_plan *p, int B, int T, const double *thr, int iters) {
    if (!p) return fail(LWS_ERR_INVALID, "null plan");
    if (B < 0 || T < 1) return fail(LWS_ERR_INVALID, "need B >= 0 and T >= 1 (got B=%d T=%d)", B, T);
    if (iters < 0) return fail(LWS_ERR_INVALID, "negative iteration count");
    if (iters > 0 && !thr) return fail(LWS_ERR_INVALID, "null threshold array");
    return LWS_OK;
}

// ---- scratch management -------------------------------------------------------------------
template <typename real>
int ensure_scratch(lws_plan *p, int B, int T, int n_thr) {
    using C = typename lws::cx<real>::type;
    const size_t Np = p->F + 2 * p->L, Tp = T + 2 * (p->Q - 1);
    int rc;
    if ((rc = p->state.ensure((size_t)B * Tp * Np * sizeof(C)))) return rc;
    if ((rc = p->amp.ensure((size_t)B * Tp * Np * sizeof(real)))) return rc;
    if ((rc = p->row_sums.ensure((size_t)B * T * sizeof(double)))) return rc;
    if ((rc = p->mean_amp.ensure((size_t)B * sizeof(double)))) return rc;
    if ((rc = p->thr_host_copy.ensure((size_t)(n_thr > 0 ? n_thr : 1) * sizeof(double)))) return rc;
    if ((rc = p->thr_scaled.ensure((size_t)B * (n_thr > 0 ? n_thr : 1) * sizeof(real)))) return rc;
    return LWS_OK;
}

int env_int(const char *name, int dflt);

void begin_timing(lws_plan *p, hipStream_t s) {
    (void)hipEventRecord(p->ev0, s);
    p->last_launches = 0;
}
void end_timing(lws_plan *p, hipStream_t s) {
    (void)hipEventRecord(p->ev1, s);
    p->timing_pending = true;
}

// One stage on the extended device buffers: thresholds -> scaled table -> update kernel.
//   mode: lws::Mode.  Assumes state/amp/mean_amp are current.
template <typename real>
int run_stage(lws_plan *p, int mode, int wsel, int B, int T, const double *thr, int iters, int LA,
              double qdiv, hipStream_t s) {
    using C = typename lws::cx<real>::type;
    if (iters <= 0) return LWS_OK;
    HIP_TRY(hipMemcpyAsync(p->thr_host_copy.p, thr, sizeof(double) * iters, hipMemcpyHostToDevice, s));
    HIP_TRY(lws::launch_scale_thresholds<real>(static_cast<const double *>(p->thr_host_copy.p),
                                              static_cast<const double *>(p->mean_amp.p),
                                              static_cast<real *>(p->thr_scaled.p), B, iters, s));
    if (mode == lws::MODE_NOFUTURE && (p->flags & LWS_NOFUTURE_Q4_COMPAT) && p->Q == 4 && p->Qp == 4)
        mode = lws::MODE_NOFUTURE_Q4_COMPAT;

    // the systolic kernel serves batch sweeps of plans it was built for (fp32, summarised weights
    // with the twiddle structure of create_weights, supported shape); everything else is generic.
    if (!p->fp64 && mode == lws::MODE_BATCH && !(p->flags & LWS_FORCE_GENERIC)) {
        if (p->sysb && p->sysb->supports(p->sys, wsel, T)) {
            int launches = 0;
            float2 *st = static_cast<float2 *>(p->state.p);
            const float *am = static_cast<const float *>(p->amp.p), *th = static_cast<const float *>(p->thr_scaled.p);
            hipError_t e = p->sysb->launch(p->sys, wsel, st, am, th, B, T, iters, s, &launches, p->ev0, p->ev1);
            p->timing_pending = true;
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "systolic launch failed: %s", hipGetErrorString(e));
            p->last_launches = launches;
            p->last_name = p->sysb->name(p->sys);
            return LWS_OK;
        }
    }

    lws::GenericArgs<real> a;
    a.state = static_cast<C *>(p->state.p);
    a.amp = static_cast<const real *>(p->amp.p);
    a.thr = static_cast<const real *>(p->thr_scaled.p);
    for (int i = 0; i < 3; ++i) a.w[i] = wset<real>(p, p->have[i] ? i : 0);
    a.wsel = wsel;
    a.F = p->F; a.T = T; a.L = p->L; a.Q = p->Q; a.Qp = p->Qp;
    a.n_thr = iters;
    a.LA = LA;
    a.M0 = 0;
    a.update = 2;  // both shipped callers pass 2 (lws.pyx:363, online_lws.cpp:160)
    a.qdiv = (real)qdiv;
    a.mode = mode;
    a.group = 1;
    // (LWS_TEAM_FIRST=1, comparison runs: the team engine before the LDS engines of the online / no-future stages)
    if ((mode == lws::MODE_ONLINE || mode == lws::MODE_NOFUTURE) && !(p->flags & LWS_FORCE_GENERIC) && env_int("LWS_TEAM_FIRST", 0) &&
        (!p->fp64 || env_int("LWS_TEAM_FP64", 0) || mode == lws::MODE_ONLINE) &&
        lws::team_supports(mode, a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr) &&
        !(mode == lws::MODE_ONLINE && lws::team_online_is_ordered(p->fp64) && !lws::team_ordered_fits(a.F, a.T, a.L, a.Q, a.LA, a.n_thr, p->fp64))) {
        const bool ordered = mode == lws::MODE_ONLINE && lws::team_online_is_ordered(p->fp64);
        begin_timing(p, s);
        hipError_t e = lws::launch_team<real>(a, B, s);
        end_timing(p, s);
        if (e != hipSuccess) return fail(LWS_ERR_HIP, "team engine launch failed: %s", hipGetErrorString(e));
        p->last_launches = 1;
        p->last_name = mode == lws::MODE_ONLINE ? (ordered ? (p->fp64 ? "team_online_ordered_fp64" : "team_online_ordered_fp32") : (p->fp64 ? "team_online_fp64" : "team_online_fp32"))
                                                : (p->fp64 ? "team_nofuture_fp64" : "team_nofuture_fp32");
        return LWS_OK;
    }
    if constexpr (std::is_same<real, float>::value) {
        // online driver: frames of the moving window live in LDS when the shape allows it
        if (mode == lws::MODE_ONLINE && !(p->flags & LWS_FORCE_GENERIC) &&
            lws::online_lds_supports(a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr, a.update, p->twiddle_all ? p->tw_P : 0, p->tw_s, p->online_tw.p != nullptr)) {
            begin_timing(p, s);
            hipError_t e = lws::launch_online_lds(a, B, p->tw_P, p->tw_s, static_cast<const float *>(p->online_tw.p), s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "online launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = "online_lds_fp32";
            return LWS_OK;
        }
    }
    if constexpr (std::is_same<real, float>::value) {
        // no-future sweeps: the last Q + 1 frames live in LDS (same results as the generic engine, bit for bit)
        if ((mode == lws::MODE_NOFUTURE || mode == lws::MODE_NOFUTURE_Q4_COMPAT) && !(p->flags & LWS_FORCE_GENERIC) &&
            lws::nofuture_lds_supports(a.F, a.T, a.L, a.Q, a.Qp, p->wperiod[a.wsel])) {
            begin_timing(p, s);
            hipError_t e = lws::launch_nofuture_lds(a, B, p->wperiod[a.wsel], s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "no-future launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = mode == lws::MODE_NOFUTURE_Q4_COMPAT ? "nofuture_lds_q4compat_fp32" : "nofuture_lds_fp32";
            return LWS_OK;
        }
    }
    if constexpr (std::is_same<real, double>::value) {
        // no-future sweeps of an fp64 plan: the LDS engine's one-lane-per-bin variant in double (the generic engine's bits)
        if ((mode == lws::MODE_NOFUTURE || mode == lws::MODE_NOFUTURE_Q4_COMPAT) && !(p->flags & LWS_FORCE_GENERIC) && !env_int("LWS_NO_ONLINE64", 0) &&
            lws::nofuture_lds64_supports(a.F, a.T, a.L, a.Q, a.Qp, p->wperiod[a.wsel])) {
            begin_timing(p, s);
            hipError_t e = lws::launch_nofuture_lds64(a, B, p->wperiod[a.wsel], s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "fp64 no-future launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = mode == lws::MODE_NOFUTURE_Q4_COMPAT ? "nofuture_lds_q4compat_fp64" : "nofuture_lds_fp64";
            return LWS_OK;
        }
    }
    if constexpr (std::is_same<real, double>::value) {
        // online driver of an fp64 plan: the frames of the moving window in LDS, every sum in the generic engine's order (same bits)
        // (Q = 8: three frames' pairs on the two waves' chain -- 1 296 ms for 256 x 500 x 257 against 831 on the team engine's order-exact
        // kernel, which gives the same bits (the generic engine's): such plans go there unless LWS_NO_TEAM_Q8=1 asks for this kernel.
        // With LWS_TEAM_FP64=1: the team engine's re-associating kernel with its window in LDS, 483 ms)
        const bool q8_team = a.Q == 8 && !env_int("LWS_NO_TEAM", 0) && !env_int("LWS_NO_TEAM_Q8", 0) && !env_int("LWS_ONLINE_SERIAL_TAPS", 0) &&
                             lws::team_supports(mode, a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr) &&
                             (env_int("LWS_TEAM_FP64", 0) ? lws::team_online_in_lds(true, a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr)
                                                          : lws::team_ordered_fits(a.F, a.T, a.L, a.Q, a.LA, a.n_thr, true));
        if (mode == lws::MODE_ONLINE && !(p->flags & LWS_FORCE_GENERIC) && !env_int("LWS_NO_ONLINE64", 0) && !q8_team &&
            lws::online64_supports(a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr, a.update)) {
            begin_timing(p, s);
            hipError_t e = lws::launch_online64(a, B, s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "fp64 online launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = lws::online64_name();
            return LWS_OK;
        }
    }
    if constexpr (std::is_same<real, double>::value) {
        // batch sweeps of an fp64 plan: the fp64 systolic engine (lws_sys64.hip) when the shape and the weights allow it.  Same
        // sweeps in the reference's order; a bin's sum is taken in another order, so results agree to rounding, not bit for bit
        // (LWS_FORCE_GENERIC keeps the order-exact engine).
        if (mode == lws::MODE_BATCH && !(p->flags & (LWS_FORCE_GENERIC | LWS_GENERIC_PLAIN_LAYOUT)) && !env_int("LWS_NO_SYS64", 0) &&
            lws::sys64_supports(a.F, a.T, a.L, a.Q, a.Qp, a.update, p->have[wsel] ? p->hostW[wsel].data() : nullptr)) {
            size_t ab = 0;
            const size_t sb = lws::sys64_bytes(B, a.F, a.T, a.Q, &ab);
            if (p->gsk_state.ensure(sb) == LWS_OK && p->gsk_amp.ensure(ab) == LWS_OK) {
                int launches = 0;
                hipError_t e = lws::launch_sys64(a, p->hostW[wsel].data(), B, p->gsk_state.p, p->gsk_amp.p, s, &launches, p->ev0, p->ev1);
                p->timing_pending = true;
                if (e != hipSuccess) return fail(LWS_ERR_HIP, "fp64 systolic launch failed: %s", hipGetErrorString(e));
                p->last_launches = launches;
                p->last_name = lws::sys64_name(a.F, a.T, a.Q);
                return LWS_OK;
            }
            p->gsk_state.release(); p->gsk_amp.release();   // no room for the skewed copy: the generic engine below
            (void)hipGetLastError();
            g_err.clear();
        }
    }
    // batch sweeps no systolic build takes (5-8 frames per stencil row above 513 bins, more than 8 frames per row, stencils of
    // half-width 6-10, fp64 plans beyond Q in {2, 4}): the band engine (lws_band.hip) when the weights have create_weights'
    // twiddle structure and a sweep slot's ring fits the LDS.  Same sweeps in the reference's order; a bin's sum in another order.
    if (mode == lws::MODE_BATCH && !(p->flags & (LWS_FORCE_GENERIC | LWS_GENERIC_PLAIN_LAYOUT)) && !env_int("LWS_NO_BAND", 0) && p->have[wsel]) {
        lws::BandPlan bp;
        if (lws::band_plan(p->fp64, B, a.F, a.T, a.L, a.Q, a.Qp, a.update, a.n_thr, p->hostW[wsel].data(), &bp)) {
            if (p->band_tab_lt[wsel] != bp.LT) {   // (once per plan and tensor: a blocking copy of a few KB)
                const std::vector<unsigned char> tab = lws::band_tables(bp, p->hostW[wsel].data());
                int rc = p->band_tab[wsel].ensure(tab.size());
                if (rc) return rc;
                HIP_TRY(hipMemcpy(p->band_tab[wsel].p, tab.data(), tab.size(), hipMemcpyHostToDevice));
                p->band_tab_lt[wsel] = bp.LT;
            }
            if (p->gsk_state.ensure(bp.state_bytes) == LWS_OK && p->gsk_amp.ensure(bp.amp_bytes) == LWS_OK) {
                int launches = 0;
                hipError_t e = lws::launch_band<real>(bp, a, p->band_tab[wsel].p, B, p->gsk_state.p, p->gsk_amp.p, s, &launches, p->ev0, p->ev1);
                p->timing_pending = true;
                if (e != hipSuccess) return fail(LWS_ERR_HIP, "band engine launch failed: %s", hipGetErrorString(e));
                p->last_launches = launches;
                p->last_name = lws::band_name(bp);
                return LWS_OK;
            }
            p->gsk_state.release(); p->gsk_amp.release();   // no room for the skewed copy: the generic engine below
            (void)hipGetLastError();
            g_err.clear();
        }
    }
    if (mode == lws::MODE_BATCH && !(p->flags & LWS_GENERIC_PLAIN_LAYOUT)) {
        // batch sweeps of the generic engine run on a time-skewed copy of the state (coalesced taps; same bits) unless
        // that copy would be unreasonably large
        size_t ab = 0;
        const size_t sb = lws::generic_skew_bytes<real>(B, p->F, T, p->L, p->Q, &ab);
        // (what the copy may take: half of what is free now plus what the plan already holds for it, 48 GiB at most)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        const size_t limit = std::min((size_t)48 << 30, free_b / 2 + p->gsk_state.cap + p->gsk_amp.cap);
        bool have_copy = sb + ab <= limit;
        if (have_copy && (p->gsk_state.ensure(sb) != LWS_OK || p->gsk_amp.ensure(ab) != LWS_OK)) {
            // no room after all: give back what was taken and run in the plain layout (same results, bit for bit)
            p->gsk_state.release(); p->gsk_amp.release();
            have_copy = false;
            (void)hipGetLastError();   // the refused hipMalloc must not be what the plain-layout launch below reports
            g_err.clear();
        }
        if (have_copy) {
            begin_timing(p, s);
            hipError_t e = lws::launch_generic_skewed<real>(a, B, p->gsk_state.p, p->gsk_amp.p, s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "generic (skewed) launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = p->fp64 ? "generic_skew_fp64" : "generic_skew_fp32";
            p->generic_stage = "batch";
            return LWS_OK;
        }
    }
    // online and no-future sweeps no LDS engine takes (more than 8 frames per stencil row, L > 5, weights without the twiddle
    // structure, frames beyond the rings): the team engine (lws_team.hip) -- the generic engine's schedule with a bin's taps spread
    // over a team of lanes.  Same sweeps in the reference's order; a bin's sum in another order.
    // (the serial-taps verification variants of the LDS engines promise the generic engine's bits: they keep falling through to it.
    // fp64 plans: the online and no-future recursions amplify the rounding of a re-associated sum by 5-10 per frame -- equally valid
    // phases, but not the reference's numbers an fp64 plan exists to reproduce.  Their online stage runs on the team engine's
    // ORDER-EXACT kernel (increments by many lanes, the sum by one, in the reference's order: the generic engine's bits); the
    // re-associating kernels, and no-future sweeps, only with LWS_TEAM_FP64=1.  fp32 plans: LWS_TEAM_ORDERED=1 selects that kernel too.)
    {
        const bool ordered = mode == lws::MODE_ONLINE && lws::team_online_is_ordered(p->fp64);
        const bool allowed = p->fp64 ? (ordered || env_int("LWS_TEAM_FP64", 0)) : true;
        if ((mode == lws::MODE_ONLINE || mode == lws::MODE_NOFUTURE) && !(p->flags & LWS_FORCE_GENERIC) && !env_int("LWS_NO_TEAM", 0) && allowed &&
            !env_int("LWS_ONLINE_SERIAL_TAPS", 0) && !env_int("LWS_NOFUTURE_SERIAL_TAPS", 0) && !(p->fp64 && env_int("LWS_NO_ONLINE64", 0)) &&
            lws::team_supports(mode, a.F, a.T, a.L, a.Q, a.Qp, a.LA, a.n_thr) &&
            (!ordered || lws::team_ordered_fits(a.F, a.T, a.L, a.Q, a.LA, a.n_thr, p->fp64))) {
            begin_timing(p, s);
            hipError_t e = lws::launch_team<real>(a, B, s);
            end_timing(p, s);
            if (e != hipSuccess) return fail(LWS_ERR_HIP, "team engine launch failed: %s", hipGetErrorString(e));
            p->last_launches = 1;
            p->last_name = mode == lws::MODE_ONLINE ? (ordered ? (p->fp64 ? "team_online_ordered_fp64" : "team_online_ordered_fp32") : (p->fp64 ? "team_online_fp64" : "team_online_fp32"))
                                                    : (p->fp64 ? "team_nofuture_fp64" : "team_nofuture_fp32");
            return LWS_OK;
        }
    }
    begin_timing(p, s);
    hipError_t e = lws::launch_generic<real>(a, B, s);
    end_timing(p, s);
    if (e != hipSuccess) return fail(LWS_ERR_HIP, "generic launch failed: %s", hipGetErrorString(e));
    p->last_launches = 1;
    p->last_name = p->fp64 ? "generic_fp64" : "generic_fp32";
    p->generic_stage = mode == lws::MODE_BATCH ? "batch" : (mode == lws::MODE_ONLINE ? "online" : "no-future");
    return LWS_OK;
}

struct StageSpec {
    int mode, wsel;
    const double *thr;
    int iters;
    int LA;
    double qdiv;
};

// prep -> stages (with pad refresh in between) -> extract, for either host (double2) or device
// (float2 / double2, in place) spectrogram buffers.
// A call that is exactly one batch stage the systolic kernel can serve, on device complex64 spectrograms: the
// spectrograms go straight into the kernel's layout and straight back (no extended buffers, no prep / extract passes).
int run_direct_batch(lws_plan *p, const float2 *in_dev, float2 *out_dev, int B, int T, const StageSpec &st, hipStream_t s) {
    int rc;
    if ((rc = p->mean_amp.ensure((size_t)B * sizeof(double)))) return rc;
    if ((rc = p->thr_host_copy.ensure((size_t)st.iters * sizeof(double)))) return rc;
    if ((rc = p->thr_scaled.ensure((size_t)B * st.iters * sizeof(float)))) return rc;
    const size_t n_part = p->sysb->io_partials(p->sys, T);
    if ((rc = p->row_sums.ensure((size_t)B * n_part * sizeof(double)))) return rc;
    double *partial = static_cast<double *>(p->row_sums.p), *mean = static_cast<double *>(p->mean_amp.p);
    hipError_t e = p->sysb->io_load(p->sys, in_dev, B, T, st.iters, partial, mean, s);
    if (e != hipSuccess) return fail(LWS_ERR_HIP, "systolic load failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpyAsync(p->thr_host_copy.p, st.thr, sizeof(double) * st.iters, hipMemcpyHostToDevice, s));
    HIP_TRY(lws::launch_scale_thresholds<float>(static_cast<const double *>(p->thr_host_copy.p), mean,
                                                static_cast<float *>(p->thr_scaled.p), B, st.iters, s));
    if (p->ev_after_load) HIP_TRY(hipEventRecord(p->ev_after_load, s));
    int launches = 0;
    const float *th = static_cast<const float *>(p->thr_scaled.p);
    e = p->sysb->io_run(p->sys, st.wsel, th, in_dev, out_dev, partial, B, T, st.iters, s, &launches, p->ev0, p->ev1);
    p->timing_pending = true;
    if (e != hipSuccess) return fail(LWS_ERR_HIP, "systolic launch failed: %s", hipGetErrorString(e));
    p->last_launches = launches;
    p->last_name = p->sysb->name(p->sys);
    return LWS_OK;
}

template <typename real, typename io_cx>
int run_pipeline(lws_plan *p, const io_cx *in_dev, io_cx *out_dev, const io_cx *orig_dev, int B,
                 int T, const StageSpec *stages, int nstages, hipStream_t s) {
    using C = typename lws::cx<real>::type;
    if constexpr (std::is_same<real, float>::value && std::is_same<io_cx, float2>::value) {
        int active = 0, which = -1;
        for (int i = 0; i < nstages; ++i)
            if (stages[i].iters > 0) { ++active; which = i; }
        if (active == 1 && stages[which].mode == lws::MODE_BATCH && !(p->flags & (LWS_FORCE_GENERIC | LWS_NO_DIRECT_IO)) &&
            p->sysb && p->sysb->supports(p->sys, stages[which].wsel, T))
            return run_direct_batch(p, in_dev, out_dev, B, T, stages[which], s);
    }
    int max_it = 1;
    for (int i = 0; i < nstages; ++i)
        if (stages[i].iters > max_it) max_it = stages[i].iters;
    int rc = ensure_scratch<real>(p, B, T, max_it);
    if (rc) return rc;
    HIP_TRY((lws::launch_prep<real, io_cx>(in_dev, static_cast<C *>(p->state.p),
                                           static_cast<real *>(p->amp.p),
                                           static_cast<double *>(p->row_sums.p),
                                           static_cast<double *>(p->mean_amp.p), B, T, p->F, p->L,
                                           p->Q, s)));
    if (p->ev_after_load) HIP_TRY(hipEventRecord(p->ev_after_load, s));
    bool dirty = false;
    for (int i = 0; i < nstages; ++i) {
        if (stages[i].iters <= 0) continue;  // "return S" of lws.pyx:219-220 / 272-273 / 332-333
        if (dirty)
            HIP_TRY(lws::launch_refresh<real>(static_cast<C *>(p->state.p),
                                              static_cast<real *>(p->amp.p),
                                              static_cast<double *>(p->row_sums.p),
                                              static_cast<double *>(p->mean_amp.p), B, T, p->F,
                                              p->L, p->Q, s));
        rc = run_stage<real>(p, stages[i].mode, stages[i].wsel, B, T, stages[i].thr, stages[i].iters,
                             stages[i].LA, stages[i].qdiv, s);
        if (rc) return rc;
        dirty = true;
    }
    HIP_TRY((lws::launch_extract<real, io_cx>(static_cast<const C *>(p->state.p), out_dev, orig_dev, B,
                                              T, p->F, p->L, p->Q, s)));
    return LWS_OK;
}

int need_weights(const lws_plan *p, const StageSpec *st, int n) {
    for (int i = 0; i < n; ++i) {
        if (st[i].iters <= 0) continue;
        if (st[i].mode == lws::MODE_ONLINE) {
            if (!(p->have[0] && p->have[1] && p->have[2]))
                return fail(LWS_ERR_INVALID, "online LWS needs W, W_ai and W_af in the plan");
        } else {
            if (st[i].wsel < 0 || st[i].wsel > 2 || !p->have[st[i].wsel])
                return fail(LWS_ERR_INVALID, "weight tensor %d is not part of this plan", st[i].wsel);
        }
    }
    return LWS_OK;
}

// After a synchronisation point: did a multi-workgroup systolic launch give up waiting (workgroups of one spectrogram
// not co-scheduled, e.g. the device was shared)?  Not an error: the call was then re-run on the device with one
// workgroup per spectrogram before it completed (lws_systolic.hip: run_kernel); the kernel name says so.
int check_systolic_flag(lws_plan *p) {
    for (lws::SystolicPlan *sp : {&p->sys}) {
        if (sp->last_nwg > 1 && sp->err_dev) {
            int flag = 0;
            HIP_TRY(hipMemcpy(&flag, sp->err_dev, sizeof(int), hipMemcpyDeviceToHost));
            sp->last_nwg = 1;
            if (flag) p->last_name = "systolic (multi-workgroup hand-over timed out: re-run with one workgroup per spectrogram)";
        }
    }
    return LWS_OK;
}

// ---- host-array entry points of an fp32 plan -----------------------------------------------------------------------------
// What a user of the drop-in calls is lws.lws(...).run_lws(numpy) (lws.pyx:495-499, python/README.md:96-100): complex128 in
// pageable host memory, complex128 out.  One pageable copy of 16 B/bin each way around the kernels costs several times the
// kernels themselves, so the batch is cut into chunks of whole spectrograms that flow through a pipeline:
//     host threads: complex128 -> complex64 into a pinned buffer   |  H2D (copy stream)  |  stages, in place (compute stream)
//     |  D2H (copy stream) into a pinned buffer  |  host threads: complex64 -> complex128 into the caller's array
// -- 8 B/bin over the bus instead of 16, both directions and the host passes overlapped with the kernels of the neighbouring
// chunks.  A bin no sweep updated comes back as the caller's complex128 value, bit for bit (the rule of launch_extract:
// "the state still equals the rounded original"), applied here on the way up.  Thresholds are scaled by the mean magnitude
// of the complex64 values, as in the *_dev entry points.
class HostWorkers {
  public:
    explicit HostWorkers(int n) {
        for (int i = 1; i < n; ++i) th_.emplace_back([this] { loop(); });
    }
    ~HostWorkers() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    // f(i) for i in [0, n), on the workers and the calling thread; returns when all are done
    void run(int n, const std::function<void(int)> &f) {
        if (n <= 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            f_ = &f; n_ = n; next_ = 0; left_ = n; ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        f_ = nullptr;
    }

  private:
    void work() {
        for (;;) {
            int i;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!f_ || next_ >= n_) return;
                i = next_++;
                f = f_;
            }
            (*f)(i);
            std::lock_guard<std::mutex> g(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *f_ = nullptr;
    int n_ = 0, next_ = 0, left_ = 0;
    unsigned gen_ = 0;
    bool stop_ = false;
};

void narrow_c128(const double *in, float *out, size_t lo, size_t hi) {   // bins [lo, hi)
    for (size_t i = 2 * lo; i < 2 * hi; ++i) out[i] = (float)in[i];
}
// the same for an input whose imaginary parts are all zero (the documented usage: run_lws(np.abs(X)), python/README.md:98-100):
// 4 bytes per bin.  Returns false -- with the output unfinished -- at the first non-zero imaginary part.
bool narrow_real(const double *in, float *out, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
        if (in[2 * i + 1] != 0.0) return false;
        out[i] = (float)in[2 * i];
    }
    return true;
}
__global__ void __launch_bounds__(256) k_expand_real(const float *__restrict__ re, float2 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_float2(re[i], 0.f);
}
void widen_c64(const float *dev, const double *orig, double *out, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
        const float vr = dev[2 * i], vi = dev[2 * i + 1];
        const double orr = orig[2 * i], oi = orig[2 * i + 1];
        const bool same = (float)orr == vr && (float)oi == vi;      // never updated: the caller's value, bit for bit
        out[2 * i] = same ? orr : (double)vr;
        out[2 * i + 1] = same ? oi : (double)vi;
    }
}

}  // namespace
namespace lws {
int usable_cpus() {
    static const int n = [] {
        int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::max(1, CPU_COUNT(&set));
        // cgroup v2: "<quota us> <period us>" or "max <period us>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us
        double quota = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            double per = 0;
            if (fscanf(f, "%31s %lf", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0) quota = atof(q) / per;
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            double q = 0, per = 0;
            if (fscanf(g, "%lf", &q) == 1 && q > 0) {
                if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                    if (fscanf(h, "%lf", &per) == 1 && per > 0) quota = q / per;
                    fclose(h);
                }
            }
            fclose(g);
        }
        if (quota >= 1.0) cpus = std::min(cpus, (int)quota);
        return std::max(1, cpus);
    }();
    return n;
}
}  // namespace lws
void lws_plan_set_host_threads(lws_plan *p, int n) { if (p) p->host_threads = n; }
namespace {
int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// spectrograms per chunk of the pipeline below (`per`: bins of one spectrogram)
// whole_device: some stage of the call runs ONE workgroup per spectrogram (no-future, online, the generic engine): a launch of
// fewer spectrograms than CUs takes as long as one of a full device's worth, so chunks hold multiples of the CU count
// (config 3, 256 spectrograms: four chunks of 64 took 259 ms, one of 256 takes 105)
int host_chunk(size_t per, int B, int n_cu, bool whole_device = false) {
    const size_t target = (size_t)std::max(1, env_int("LWS_HOST_CHUNK_BINS", 16 << 20));
    if (per * (size_t)B <= target + target / 2) return B;
    int bc = (int)std::min<size_t>((size_t)B, std::max<size_t>(1, (target + per / 2) / per));
    if (whole_device && n_cu > 1 && !env_int("LWS_HOST_CHUNK_EXACT", 0)) {
        bc = std::min(B, std::max(n_cu, bc - bc % n_cu));
        // ... but the pipeline pins four host buffers of a chunk each (and holds two on the device): long spectrograms are cut
        // below a device's worth (half, a quarter, ... of the CUs busy in the one-workgroup stages) rather than pin tens of GB
        const size_t pin_limit = (size_t)std::max(1, env_int("LWS_HOST_PIN_MB", 2048)) << 20;
        while (bc > 1 && (size_t)bc * per * sizeof(float2) > pin_limit) bc = (bc + 1) / 2;
        return bc;
    }
    // a launch of fewer spectrograms than CUs gives each floor(CUs / spectrograms) workgroups (lws_systolic.hip: prepare):
    // 65 spectrograms on 256 CUs keep 195 of them busy, 64 all of them -- round to a divisor / multiple of the CU count
    if (n_cu > 1 && !env_int("LWS_HOST_CHUNK_EXACT", 0)) {
        if (bc >= n_cu) bc -= bc % n_cu;
        // a batch of several devices' worth: chunks of one device's worth, one workgroup per spectrogram -- the launches then run at
        // the whole-batch rate (no passes shared out between workgroups) and only the first upload / last download are exposed
        // (1024 x 500 x 513, round 5: 211 ms in chunks of 64, 154 in chunks of 256, 131 device-resident)
        else if (B >= 2 * n_cu && (size_t)n_cu * per * sizeof(float2) <= ((size_t)std::max(1, env_int("LWS_HOST_PIN_MB", 2048)) << 20)) bc = n_cu;
        else bc = std::max(1, n_cu / ((n_cu + bc - 1) / bc));
    }
    return std::min(bc, B);
}
int cu_count(int device) {
    int n = 0;
    return hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess ? n : 0;
}

// (Tried and dropped: four child plans with a quarter of the CUs each, working on a chunk each at the same time -- one
// workgroup per spectrogram in every launch, a chunk's kernels starting when ITS upload is done.  Kernels of different
// streams do overlap -- 2 x 64 spectrograms on two streams take the time of one, 33 ms -- but a process gets four hardware
// queues by default and twelve streams share them: 170 ms instead of 61; with one stream per lane the gain over this
// pipeline is bounded by ~10 %, the exposed first upload and last download being what they are.)
// the stream `s` is about to use the plan's scratch: after whatever the plan enqueued last
int order_after_plan_work(lws_plan *p, hipStream_t s) {
    if (p->busy) HIP_TRY(hipStreamWaitEvent(s, p->ev_busy, 0));
    return LWS_OK;
}
// ... and `s` now carries the plan's latest work
int note_plan_work(lws_plan *p, hipStream_t s) {
    if (!p->ev_busy) HIP_TRY(hipEventCreateWithFlags(&p->ev_busy, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(p->ev_busy, s));
    p->busy = true;
    return LWS_OK;
}

int run_host_monolithic(lws_plan *p, const double *S_in, double *S_out, int B, int T, const StageSpec *st, int n);

int run_host_pipelined(lws_plan *p, const double *S_in, double *S_out, int B, int T, const StageSpec *st, int n) {
    const size_t per = (size_t)T * p->F;                       // bins of a spectrogram
    const size_t total = per * (size_t)B;
    // chunks of whole spectrograms, ~16M bins each (128 MB of complex64; measured on 256 x 500 x 513: 8M 65 ms, 16M 60 ms,
    // 32M 72 ms): long enough for the kernels to fill the device -- a launch of 32 spectrograms takes 6.7 ms, of 64 10.5, of
    // 256 33.9: fewer spectrograms than CUs run several workgroups each, 70-85 % as efficient -- short enough for the first
    // upload and the last download, which nothing overlaps, to be a small part of the call
    bool whole_device = false;
    for (int i = 0; i < n; ++i)
        if (st[i].iters > 0 && (st[i].mode != lws::MODE_BATCH || (p->flags & LWS_FORCE_GENERIC) || !p->sysb || !p->sysb->supports(p->sys, st[i].wsel, T)))
            whole_device = true;
    const int Bc = host_chunk(per, B, cu_count(p->device), whole_device);
    // chunk c = spectrograms [cs[c], cs[c + 1]).  The first upload and the last download have nothing to overlap with: when the
    // batch is cut at all, the first chunk is half a chunk (the device starts after half the narrowing and half the copy), and the
    // remainder that leaves at the end is half a chunk too.
    std::vector<int> cs{0};
    if (Bc < B && Bc >= 2 && !whole_device && env_int("LWS_HOST_HALF_FIRST", 1)) cs.push_back(Bc / 2);
    while (cs.back() < B) cs.push_back(std::min(B, cs.back() + Bc));
    const int nch = (int)cs.size() - 1;
    // conversion threads: up to 32 (more gain nothing: the passes are memory-bound), at most the CPUs the process can use at once
    // (lws_multi_* plans: their share of them -- eight plans of 32 threads on a 16-CPU quota only fight each other)
    int nthreads = env_int("LWS_HOST_THREADS", std::min(32, p->host_threads > 0 ? p->host_threads : lws::usable_cpus()));
    if (total < ((size_t)1 << 20)) nthreads = 1;
    nthreads = std::max(1, std::min(nthreads, 64));
    HIP_TRY(hipSetDevice(p->device));
    HostPipe &hp = p->pipe;
    int rc = hp.ensure((size_t)Bc * per * sizeof(float2), nch > 1 ? 2 : 1);
    if (rc == LWS_ERR_NOMEM) {
        // no room for the pinned staging buffers (a one-workgroup-per-spectrogram stage makes chunks of at least a device's
        // worth of spectrograms): the unpipelined path needs no pinned memory
        hp.release();
        (void)hipGetLastError();
        g_err.clear();
        return run_host_monolithic(p, S_in, S_out, B, T, st, n);
    }
    if (rc) return rc;
    if ((rc = order_after_plan_work(p, hp.s_comp))) return rc;   // (*_dev calls enqueued earlier use the same scratch)
    if (p->host_pool && p->host_pool_n != nthreads) { delete static_cast<HostWorkers *>(p->host_pool); p->host_pool = nullptr; }
    if (!p->host_pool) { p->host_pool = new HostWorkers(nthreads); p->host_pool_n = nthreads; }
    HostWorkers &pool = *static_cast<HostWorkers *>(p->host_pool);
    const int slices = pool.size() == 1 ? 1 : 4 * pool.size();
    auto chunk_bins = [&](int c) { return (size_t)(cs[c + 1] - cs[c]) * per; };
    // Real-valued input (magnitudes: the documented usage run_lws(np.abs(X)), python/README.md:98-100) goes up as 4 bytes per bin
    // and is expanded on the device.  Decided chunk by chunk while narrowing: the pass that reads every element anyway stops at
    // the first non-zero imaginary part and the chunk is narrowed again as complex (an input that starts real and turns complex
    // pays for that once; a complex input fails the probe below and never tries).
    bool real_mode = env_int("LWS_HOST_REAL", 1) != 0;
    for (size_t i = 0; i < std::min<size_t>(total, 256) && real_mode; ++i) real_mode = S_in[2 * i + 1] == 0.0;
    std::vector<char> chunk_real(nch, 0);
    // LWS_HOST_TRACE=1: where the call's wall time goes, on stderr (ms since the call began)
    const bool trace = env_int("LWS_HOST_TRACE", 0) != 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto mark = [&](const char *what, int c) {
        if (trace) fprintf(stderr, "[lws host] %7.2f ms  %s %d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), what, c);
    };
    // host passes: narrow chunk cu (if any) into up[cu & 1] and widen chunk cd (if any) from down[cd & 1], together
    // ... of the bins [ulo, uhi) of chunk cu and [dlo, dhi) of chunk cd
    // (returns false if the range was to be narrowed as real values and is not real: nothing usable was written for it)
    auto host_ranges = [&](int cu, size_t ulo, size_t uhi, bool as_real, int cd, size_t dlo, size_t dhi) -> bool {
        const int nu = uhi > ulo ? slices : 0, nd = dhi > dlo ? slices : 0;
        std::atomic<bool> is_real{true};
        pool.run(nu + nd, [&](int i) {
            if (i < nu) {
                const size_t nb = uhi - ulo, lo = ulo + nb * i / slices, hi = ulo + nb * (i + 1) / slices, off = (size_t)cs[cu] * per;
                if (!as_real) narrow_c128(S_in + 2 * off, static_cast<float *>(hp.up[cu & 1]), lo, hi);
                else if (is_real.load(std::memory_order_relaxed) && !narrow_real(S_in + 2 * off, static_cast<float *>(hp.up[cu & 1]), lo, hi))
                    is_real.store(false, std::memory_order_relaxed);
            } else {
                const int k = i - nu;
                const size_t nb = dhi - dlo, lo = dlo + nb * k / slices, hi = dlo + nb * (k + 1) / slices, off = (size_t)cs[cd] * per;
                widen_c64(static_cast<const float *>(hp.down[cd & 1]), S_in + 2 * off, S_out + 2 * off, lo, hi);
            }
        });
        return is_real.load();
    };
    auto host_pass = [&](int cu, int cd) {
        const size_t un = (cu >= 0 && cu < nch) ? chunk_bins(cu) : 0, dn = (cd >= 0 && cd < nch) ? chunk_bins(cd) : 0;
        if (host_ranges(cu, 0, un, real_mode, cd, 0, dn)) {
            if (un) chunk_real[cu] = real_mode;
        } else {                       // chunk cu is not real-valued: again, as complex (the widening above is done)
            real_mode = false;
            host_ranges(cu, 0, un, false, -1, 0, 0);
        }
    };
    constexpr int NP = HostPipe::PIECES;
    auto piece = [&](int c, int q) { return chunk_bins(c) * (size_t)q / NP; };   // first bin of piece q of chunk c
    hipEvent_t tr0[16], tr1[16];   // (trace only) around the stages of each chunk, on the compute stream
    mark("pool up, chunks:", nch);
    // the first chunk goes up in pieces: piece q + 1 is narrowed while piece q is on the bus
    for (int q = 0; q < NP; ++q) {
        const size_t lo = piece(0, q), hi = piece(0, q + 1);
        if (!host_ranges(0, lo, hi, real_mode, -1, 0, 0)) {   // not real-valued after all: the chunk again from its first piece, as complex
            real_mode = false;
            q = -1;
            continue;
        }
        if (hi <= lo) continue;
        if (real_mode) HIP_TRY(hipMemcpyAsync(static_cast<float *>(hp.io_real[0].p) + lo, static_cast<const float *>(hp.up[0]) + lo, (hi - lo) * sizeof(float), hipMemcpyHostToDevice, hp.s_up));
        else HIP_TRY(hipMemcpyAsync(static_cast<float2 *>(hp.io[0].p) + lo, static_cast<const float2 *>(hp.up[0]) + lo, (hi - lo) * sizeof(float2), hipMemcpyHostToDevice, hp.s_up));
    }
    chunk_real[0] = real_mode;
    mark("narrowed and on its way", 0);
    // D2H of chunk c: a blit kernel of the runtime that fills the device.  Started when chunk c is done it would run against
    // the light first kernels of chunk c+1 (memsets, layout pass, mean: 0.2 ms alone, 2.5 ms beside it); so it waits for
    // those as well and runs beside chunk c+1's update kernel instead, which leaves it the wave slots it needs.
    auto enqueue_down = [&](int c, bool after_next_load) -> int {
        const int slot = c & 1;
        HIP_TRY(hipStreamWaitEvent(hp.s_down, hp.ev_comp[slot], 0));
        if (after_next_load) HIP_TRY(hipStreamWaitEvent(hp.s_down, hp.ev_load[(c + 1) & 1], 0));
        HIP_TRY(hipMemcpyAsync(hp.down[slot], hp.io[slot].p, chunk_bins(c) * sizeof(float2), hipMemcpyDeviceToHost, hp.s_down));
        HIP_TRY(hipEventRecord(hp.ev_down[slot], hp.s_down));
        return LWS_OK;
    };
    for (int c = 0; c < nch; ++c) {
        const int slot = c & 1, bc = cs[c + 1] - cs[c];
        const size_t bytes = chunk_bins(c) * sizeof(float2);
        float2 *io = static_cast<float2 *>(hp.io[slot].p);
        if (c >= 2) HIP_TRY(hipStreamWaitEvent(hp.s_up, hp.ev_down[slot], 0));        // chunk c-2 has left this device buffer
        if (c >= 1) HIP_TRY(hipMemcpyAsync(chunk_real[c] ? hp.io_real[slot].p : static_cast<void *>(io), hp.up[slot], chunk_real[c] ? bytes / 2 : bytes,
                                           hipMemcpyHostToDevice, hp.s_up));   // (chunk 0: above)
        HIP_TRY(hipEventRecord(hp.ev_up[slot], hp.s_up));
        HIP_TRY(hipStreamWaitEvent(hp.s_comp, hp.ev_up[slot], 0));
        if (chunk_real[c]) {   // (ordered behind chunk c-2's download through ev_up: the copy stream waited for it)
            const size_t nb = chunk_bins(c);
            hipLaunchKernelGGL(k_expand_real, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, hp.s_comp, static_cast<const float *>(hp.io_real[slot].p), io, nb);
            HIP_TRY(hipGetLastError());
        }
        if (trace && c < 16) { HIP_TRY(hipEventCreate(&tr0[c])); HIP_TRY(hipEventCreate(&tr1[c])); HIP_TRY(hipEventRecord(tr0[c], hp.s_comp)); }
        p->ev_after_load = hp.ev_load[slot];
        rc = run_pipeline<float, float2>(p, io, io, io, bc, T, st, n, hp.s_comp);
        p->ev_after_load = nullptr;
        if (rc) { (void)hipDeviceSynchronize(); return rc; }
        if (trace && c < 16) HIP_TRY(hipEventRecord(tr1[c], hp.s_comp));
        HIP_TRY(hipEventRecord(hp.ev_comp[slot], hp.s_comp));
        if (c >= 1 && (rc = enqueue_down(c - 1, true))) { (void)hipDeviceSynchronize(); return rc; }
        if (c == nch - 1) {   // the last chunk comes down in pieces: piece q is widened while piece q + 1 is on the bus
            HIP_TRY(hipStreamWaitEvent(hp.s_down, hp.ev_comp[slot], 0));
            for (int q = 0; q < NP; ++q) {
                const size_t lo = piece(c, q), hi = piece(c, q + 1);
                if (hi > lo) HIP_TRY(hipMemcpyAsync(static_cast<float2 *>(hp.down[slot]) + lo, io + lo, (hi - lo) * sizeof(float2), hipMemcpyDeviceToHost, hp.s_down));
                HIP_TRY(hipEventRecord(hp.ev_piece[q], hp.s_down));
            }
            HIP_TRY(hipEventRecord(hp.ev_down[slot], hp.s_down));
        }
        // while the device works on chunk c: the next chunk on its way up (its pinned buffer is free once chunk c-1 has been
        // uploaded), the previous one on its way out (once it has arrived)
        mark("enqueued", c);
        // (after the second chunk is on its way, so that the device never waits for this: it takes ~10 ms for 1 GB)
        if (c == std::min(1, nch - 1) && static_cast<const void *>(S_out) != static_cast<const void *>(S_in)) {
            // A result array fresh from the allocator has no pages yet: 256K first-touch faults for 1 GB, which the widening
            // passes would take one by one on the critical path (and 32 threads faulting on one address space queue up in the
            // kernel: 8.5 ms for the last chunk alone).  Populate it now, eight ways, while the device works on the first chunks.
            char *lo = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(S_out) + 4095) & ~(uintptr_t)4095);
            char *hi = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(S_out) + total * 16) & ~(uintptr_t)4095);
            const int ways = std::min(8, pool.size());
            std::atomic<bool> populate_fallback{false};
            if (hi > lo && env_int("LWS_HOST_PREFAULT", 1))
                pool.run(ways, [&](int i) {
                    const size_t pages = (size_t)(hi - lo) >> 12, a = pages * i / ways, b = pages * (i + 1) / ways;
                    if (b > a && madvise(lo + (a << 12), (b - a) << 12, LWS_MADV_POPULATE_WRITE) != 0) {
                        // (older kernel, or a mapping it refuses) touch the pages instead: the array is all output -- nothing of
                        // it is read before the widening passes overwrite every element
                        for (size_t pg = a; pg < b; ++pg) *reinterpret_cast<volatile char *>(lo + (pg << 12)) = 0;
                        populate_fallback.store(true, std::memory_order_relaxed);
                    }
                });
            mark(populate_fallback.load() ? "result pages touched (madvise refused)" : "result pages populated", 0);
        }
        if (c + 1 < nch && c >= 1) HIP_TRY(hipEventSynchronize(hp.ev_up[(c + 1) & 1]));
        if (c >= 1) HIP_TRY(hipEventSynchronize(hp.ev_down[(c - 1) & 1]));
        mark("arrived", c - 1);
        host_pass(c + 1, c - 1);
        mark("host pass done: narrowed c+1, widened c-1; c =", c);
    }
    for (int q = 0; q < NP; ++q) {
        HIP_TRY(hipEventSynchronize(hp.ev_piece[q]));
        host_ranges(-1, 0, 0, false, nch - 1, piece(nch - 1, q), piece(nch - 1, q + 1));
    }
    mark("arrived and widened", nch - 1);
    if (trace) {
        for (int c = 0; c < nch && c < 16; ++c) {
            float ms = 0.f, since = 0.f;
            (void)hipEventElapsedTime(&ms, tr0[c], tr1[c]);
            (void)hipEventElapsedTime(&since, tr0[0], tr0[c]);
            fprintf(stderr, "[lws host] chunk %d: stages %.2f ms on the device, started %.2f ms after chunk 0\n", c, ms, since);
        }
        for (int c = 0; c < nch && c < 16; ++c) { (void)hipEventDestroy(tr0[c]); (void)hipEventDestroy(tr1[c]); }
    }
    p->busy = false;     // every chunk has come down: nothing of this plan is in flight
    return check_systolic_flag(p);
}

// host complex128 in/out
int run_host(lws_plan *p, const double *S_in, double *S_out, int B, int T, const StageSpec *st, int n) {
    if (!S_in || !S_out) return fail(LWS_ERR_INVALID, "null spectrogram pointer");
    int rc = need_weights(p, st, n);
    if (rc) return rc;
    const size_t count = (size_t)B * T * p->F;
    bool any = false;
    for (int i = 0; i < n; ++i) any |= st[i].iters > 0;
    if (!any || B == 0) {
        if (S_out != S_in) memcpy(S_out, S_in, count * 2 * sizeof(double));
        return LWS_OK;
    }
    if (!p->fp64 && !env_int("LWS_HOST_MONOLITHIC", 0)) return run_host_pipelined(p, S_in, S_out, B, T, st, n);
    return run_host_monolithic(p, S_in, S_out, B, T, st, n);
}

// fp64 plans (the reference's arithmetic; the parity anchor), and fp32 plans without room for pinned staging: one complex128
// copy each way around the stages
int run_host_monolithic(lws_plan *p, const double *S_in, double *S_out, int B, int T, const StageSpec *st, int n) {
    const size_t count = (size_t)B * T * p->F;
    int rc;
    HIP_TRY(hipSetDevice(p->device));
    if ((rc = p->stage.ensure(count * sizeof(double2)))) return rc;
    hipStream_t s = nullptr;
    if ((rc = order_after_plan_work(p, s))) return rc;
    HIP_TRY(hipMemcpyAsync(p->stage.p, S_in, count * sizeof(double2), hipMemcpyHostToDevice, s));
    double2 *io = static_cast<double2 *>(p->stage.p);
    if (p->fp64) rc = run_pipeline<double, double2>(p, io, io, io, B, T, st, n, s);
    else rc = run_pipeline<float, double2>(p, io, io, io, B, T, st, n, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(S_out, p->stage.p, count * sizeof(double2), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    p->busy = false;
    return check_systolic_flag(p);
}

// device in place: complex64 for fp32 plans, complex128 for fp64 plans
int run_dev(lws_plan *p, void *S_dev, int B, int T, const StageSpec *st, int n, void *stream) {
    if (!S_dev) return fail(LWS_ERR_INVALID, "null device pointer");
    int rc = need_weights(p, st, n);
    if (rc) return rc;
    bool any = false;
    for (int i = 0; i < n; ++i) any |= st[i].iters > 0;
    if (!any || B == 0) return LWS_OK;
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = order_after_plan_work(p, s))) return rc;   // (a no-op on the stream that carries the previous call)
    if (p->fp64) {
        double2 *io = static_cast<double2 *>(S_dev);
        rc = run_pipeline<double, double2>(p, io, io, io, B, T, st, n, s);
    } else {
        float2 *io = static_cast<float2 *>(S_dev);
        rc = run_pipeline<float, float2>(p, io, io, io, B, T, st, n, s);
    }
    if (rc) return rc;
    return note_plan_work(p, s);
}

}  // namespace

extern "C" {

int lws_hip_version(void) { return 100; }  // 0.1.0

const char *lws_last_error(void) { return g_err.c_str(); }

typedef float vec4f __attribute__((ext_vector_type(4)));
// one workgroup moves 4 x 256 x 16 B with non-temporal accesses; no grid-stride loop -- measured 6.0-6.5 TB/s on MI355X
// against 4.4-4.9 TB/s for a grid-stride loop and 5.0 TB/s for hipMemcpyDtoD (scratch/copy_ubench.hip)
__global__ void k_stream_copy(vec4f *__restrict__ dst, const vec4f *__restrict__ src, size_t n) {
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    vec4f v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n) v[u] = __builtin_nontemporal_load(src + base + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (base + u * 256 < n) __builtin_nontemporal_store(v[u], dst + base + u * 256);
}

int lws_stream_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    if (!dst_dev || !src_dev || (bytes & 15)) {
        return lws::set_error(LWS_ERR_INVALID, "lws_stream_copy: null pointer or size not a multiple of 16");
    }
    if (bytes == 0) return LWS_OK;
    const size_t n = bytes / 16;
    const size_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (vec4f *)dst_dev, (const vec4f *)src_dev, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lws::set_error(LWS_ERR_HIP, "lws_stream_copy: %s", hipGetErrorString(e));
    return LWS_OK;
}

int lws_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lws_plan_create(lws_plan **plan, int device, int F, int L, int Q, int Qp, const double *W,
                    const double *W_ai, const double *W_af, unsigned flags) {
    if (!plan) return fail(LWS_ERR_INVALID, "null plan pointer");
    *plan = nullptr;
    if (F < 3 || (F % 2) == 0)
        return fail(LWS_ERR_INVALID,
                    "Please only include non-negative frequencies in the input spectrogram. (F=%d must be odd)", F);
    if (L < 1 || Q < 1) return fail(LWS_ERR_INVALID, "need L >= 1 and Q >= 1 (got L=%d Q=%d)", L, Q);
    if (L > F - 2) return fail(LWS_ERR_INVALID, "L=%d is too large for F=%d bins", L, F);
    if (Qp != Q && Qp != 2 * (F - 1))
        return fail(LWS_ERR_INVALID,
                    "weight tensor has %d rows; expected Q=%d (summarised) or N=2(F-1)=%d (general)", Qp, Q,
                    2 * (F - 1));
    if (!W) return fail(LWS_ERR_INVALID, "null weight tensor");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(LWS_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    lws_plan *p = new (std::nothrow) lws_plan();
    if (!p) return fail(LWS_ERR_NOMEM, "out of host memory");
    p->device = device;
    p->F = F; p->L = L; p->Q = Q; p->Qp = Qp;
    p->flags = flags;
    p->fp64 = (flags & LWS_PRECISION_FP64) != 0;
    if (p->fp64 && (flags & LWS_STORAGE_FP16)) {
        delete p;
        return fail(LWS_ERR_INVALID, "LWS_STORAGE_FP16 is a storage mode of the fp32 engine; it cannot be combined with LWS_PRECISION_FP64");
    }
    int rc = LWS_OK;
    const double *src[3] = {W, W_ai, W_af};
    for (int i = 0; i < 3 && rc == LWS_OK; ++i) {
        if (!src[i]) continue;
        rc = p->fp64 ? upload_weights<double>(p, i, src[i]) : upload_weights<float>(p, i, src[i]);
    }
    if (rc == LWS_OK) {
        if (hipEventCreate(&p->ev0) != hipSuccess || hipEventCreate(&p->ev1) != hipSuccess)
            rc = fail(LWS_ERR_HIP, "hipEventCreate failed");
    }
    for (int i = 0; i < 3; ++i)
        p->wperiod[i] = !p->have[i] ? 0 : (Qp == Q ? Q : lws::weights_row_period(p->hostW[i].data(), Qp, Q, L, 256));
    p->twiddle_all = rc == LWS_OK && p->have[0] && p->have[1] && p->have[2];
    p->tw_P = 0;
    for (int i = 0; i < 3 && p->twiddle_all; ++i) {
        int P = 0, sg = 0;      // (P = 0: a tensor without neighbour-frame weights fits any twiddle)
        p->twiddle_all = lws::weights_twiddle(p->hostW[i].data(), Q, Qp, L, 512, &P, &sg) && (P == 0 || p->tw_P == 0 || (P == p->tw_P && sg == p->tw_s));
        if (P > 0) { p->tw_P = P; p->tw_s = sg; }
    }
    if (p->twiddle_all && p->tw_P == 0) { p->tw_P = Q; p->tw_s = 1; }
    if (rc == LWS_OK && p->twiddle_all && !p->fp64 && !lws::online_static_twiddles(Q, p->tw_P, p->tw_s) && Q <= 8) {
        std::vector<float> tab((size_t)(p->tw_P + 3) * (Q <= 4 ? 8 : 16));
        lws::online_twiddle_table(p->tw_P, p->tw_s, Q, tab.data());
        if ((rc = p->online_tw.ensure(tab.size() * sizeof(float))) == LWS_OK &&
            hipMemcpy(p->online_tw.p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(LWS_ERR_HIP, "twiddle table upload failed");
    }
    // (LWS_NO_SYSTOLIC=1, read here: no systolic build -- comparison runs of the engines behind them)
    if (rc == LWS_OK && !p->fp64 && !(flags & LWS_FORCE_GENERIC) && !env_int("LWS_NO_SYSTOLIC", 0)) {
        const double *hw[3] = {p->have[0] ? p->hostW[0].data() : nullptr,
                               p->have[1] ? p->hostW[1].data() : nullptr,
                               p->have[2] ? p->hostW[2].data() : nullptr};
        const bool h16 = (flags & LWS_STORAGE_FP16) != 0;
        hipError_t e = hipSuccess;
        // the first build that takes the shape: short frames (<= 129 / 257 bins: four / two sweep slots per wave), up to 513 bins,
        // Q = 8, up to 1025 bins.  LWS_SYSTOLIC_NO_SHORT=1 skips the short-frame builds (comparison runs)
        const bool no_short = env_int("LWS_SYSTOLIC_NO_SHORT", 0) != 0;
        for (const lws::SystolicBuild *b : {&lws::quarter_q2::systolic_entry(), &lws::quarter::systolic_entry(), &lws::half_q2::systolic_entry(), &lws::half::systolic_entry(),
                                            &lws::q2::systolic_entry(), &lws::systolic_entry(),
                                            &lws::q8::systolic_entry(), &lws::wide_q2::systolic_entry(), &lws::wide::systolic_entry(), &lws::xwide::systolic_entry(), &lws::l7::systolic_entry(),
                                            // ... then the table-twiddle builds: Q = 3, and general weights of a hop that does not divide the frame
                                            &lws::tw_half::systolic_entry(), &lws::tw::systolic_entry(), &lws::tw_wide::systolic_entry(),
                                            // (exactly 5 / 6 frames per stencil row: the builds with their own ring depth first; LWS_SYSTOLIC_NO_TWQ=1 skips them -- comparison runs)
                                            &lws::tw_q5::systolic_entry(), &lws::tw_q6::systolic_entry(), &lws::tw_q8::systolic_entry()}) {
            const bool is_short = b == &lws::quarter::systolic_entry() || b == &lws::half::systolic_entry() || b == &lws::quarter_q2::systolic_entry() ||
                                  b == &lws::half_q2::systolic_entry() || b == &lws::tw_half::systolic_entry();
            const bool is_twq = b == &lws::tw_q5::systolic_entry() || b == &lws::tw_q6::systolic_entry();
            const bool is_tw = is_twq || b == &lws::tw_half::systolic_entry() || b == &lws::tw::systolic_entry() || b == &lws::tw_wide::systolic_entry() || b == &lws::tw_q8::systolic_entry();
            if (is_twq && env_int("LWS_SYSTOLIC_NO_TWQ", 0)) continue;
            if (is_tw && env_int("LWS_SYSTOLIC_NO_TW", 0)) continue;                                 // (comparison runs)
            const bool is_r16 = b == &lws::q2::systolic_entry() || b == &lws::wide_q2::systolic_entry() || b == &lws::quarter_q2::systolic_entry() ||
                                b == &lws::half_q2::systolic_entry();
            if ((no_short && is_short) || (is_r16 && env_int("LWS_SYSTOLIC_NO_R16", 0))) continue;   // (comparison runs)
            if ((e = b->build(p->sys, F, L, Q, Qp, hw, h16)) != hipSuccess) break;
            // the build must take the tensor batch sweeps normally run on -- W, the first one present -- : a build that only takes
            // another of the plan's tensors (W_ai of a hop above half the frame has no neighbour-frame weights and fits any
            // twiddle) would leave the batch stage on the generic engine
            const int first = p->have[0] ? 0 : (p->have[1] ? 1 : 2);
            if (p->sys.ok[first]) { p->sysb = b; break; }
            if (p->sys.ok[0] || p->sys.ok[1] || p->sys.ok[2]) b->release(p->sys);
        }
        if (e != hipSuccess) rc = fail(LWS_ERR_HIP, "systolic table upload failed: %s", hipGetErrorString(e));
    }
    if (rc != LWS_OK) {
        lws_plan_destroy(p);
        return rc;
    }
    *plan = p;
    return LWS_OK;
}

void lws_plan_destroy(lws_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (int i = 0; i < 3; ++i) { p->w[i].release(); p->wflag[i].release(); }
    p->state.release(); p->amp.release(); p->row_sums.release(); p->mean_amp.release();
    p->thr_host_copy.release(); p->thr_scaled.release(); p->stage.release();
    p->resid_rows.release(); p->resid_out.release(); p->resid_sum.release(); p->gsk_state.release(); p->gsk_amp.release(); p->online_tw.release();
    for (int i = 0; i < 3; ++i) p->band_tab[i].release();
    p->pipe.release();
    delete static_cast<HostWorkers *>(p->host_pool);
    p->host_pool = nullptr;
    (p->sysb ? p->sysb : &lws::systolic_entry())->release(p->sys);
    if (p->ev_busy) (void)hipEventDestroy(p->ev_busy);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    delete p;
}

int lws_batch_lws(lws_plan *p, int wsel, const double *S_in, double *S_out, int B, int T,
                  const double *thresholds, int iters) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    StageSpec st{lws::MODE_BATCH, wsel, thresholds, iters, 0, (double)p->Q};
    return run_host(p, S_in, S_out, B, T, &st, 1);
}

int lws_nofuture_lws(lws_plan *p, int wsel, const double *S_in, double *S_out, int B, int T,
                     const double *thresholds, int iters) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    StageSpec st{lws::MODE_NOFUTURE, wsel, thresholds, iters, 0, (double)p->Q};
    return run_host(p, S_in, S_out, B, T, &st, 1);
}

int lws_online_lws(lws_plan *p, const double *S_in, double *S_out, int B, int T,
                   const double *thresholds, int iters, int LA, double qdiv) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    if (LA < 0) return fail(LWS_ERR_INVALID, "negative look-ahead");
    StageSpec st{lws::MODE_ONLINE, 0, thresholds, iters, LA, qdiv};
    return run_host(p, S_in, S_out, B, T, &st, 1);
}

int lws_run_lws(lws_plan *p, const double *S_in, double *S_out, int B, int T,
                const double *thr_nofuture, int it_nofuture, const double *thr_online, int it_online,
                int LA, double qdiv, const double *thr_batch, int it_batch) {
    int rc = check_common(p, B, T, thr_nofuture, it_nofuture);
    if (!rc) rc = check_common(p, B, T, thr_online, it_online);
    if (!rc) rc = check_common(p, B, T, thr_batch, it_batch);
    if (rc) return rc;
    if (LA < 0) return fail(LWS_ERR_INVALID, "negative look-ahead");
    StageSpec st[3] = {{lws::MODE_NOFUTURE, LWS_W_AI, thr_nofuture, it_nofuture, 0, (double)p->Q},
                       {lws::MODE_ONLINE, 0, thr_online, it_online, LA, qdiv},
                       {lws::MODE_BATCH, LWS_W, thr_batch, it_batch, 0, (double)p->Q}};
    return run_host(p, S_in, S_out, B, T, st, 3);
}

int lws_batch_lws_dev(lws_plan *p, int wsel, void *S_dev, int B, int T, const double *thresholds,
                      int iters, void *stream) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    StageSpec st{lws::MODE_BATCH, wsel, thresholds, iters, 0, (double)p->Q};
    return run_dev(p, S_dev, B, T, &st, 1, stream);
}

int lws_nofuture_lws_dev(lws_plan *p, int wsel, void *S_dev, int B, int T, const double *thresholds,
                         int iters, void *stream) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    StageSpec st{lws::MODE_NOFUTURE, wsel, thresholds, iters, 0, (double)p->Q};
    return run_dev(p, S_dev, B, T, &st, 1, stream);
}

int lws_online_lws_dev(lws_plan *p, void *S_dev, int B, int T, const double *thresholds, int iters,
                       int LA, double qdiv, void *stream) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    if (LA < 0) return fail(LWS_ERR_INVALID, "negative look-ahead");
    StageSpec st{lws::MODE_ONLINE, 0, thresholds, iters, LA, qdiv};
    return run_dev(p, S_dev, B, T, &st, 1, stream);
}

int lws_run_lws_dev(lws_plan *p, void *S_dev, int B, int T, const double *thr_nofuture, int it_nofuture,
                    const double *thr_online, int it_online, int LA, double qdiv, const double *thr_batch, int it_batch,
                    void *stream) {
    int rc = check_common(p, B, T, thr_nofuture, it_nofuture);
    if (!rc) rc = check_common(p, B, T, thr_online, it_online);
    if (!rc) rc = check_common(p, B, T, thr_batch, it_batch);
    if (rc) return rc;
    if (LA < 0) return fail(LWS_ERR_INVALID, "negative look-ahead");
    StageSpec st[3] = {{lws::MODE_NOFUTURE, LWS_W_AI, thr_nofuture, it_nofuture, 0, (double)p->Q},
                       {lws::MODE_ONLINE, 0, thr_online, it_online, LA, qdiv},
                       {lws::MODE_BATCH, LWS_W, thr_batch, it_batch, 0, (double)p->Q}};
    return run_dev(p, S_dev, B, T, st, 3, stream);
}

int lws_plan_reserve(lws_plan *p, int B, int T, int max_iters) {
    if (!p) return fail(LWS_ERR_INVALID, "null plan");
    if (B < 1 || T < 1 || max_iters < 0) return fail(LWS_ERR_INVALID, "need B >= 1, T >= 1, max_iters >= 0");
    HIP_TRY(hipSetDevice(p->device));
    int rc = p->fp64 ? ensure_scratch<double>(p, B, T, max_iters) : ensure_scratch<float>(p, B, T, max_iters);
    if (rc) return rc;
    const size_t count = (size_t)B * T * p->F;
    if ((rc = p->stage.ensure(count * sizeof(double2)))) return rc;          // complex128 staging (fp64 plans' host entry points, lws_residual)
    if (!p->fp64) {   // host entry points of an fp32 plan: pinned buffers, chunk buffers and streams of the pipeline, so that the
                      // first call does not pay for them (hipHostMalloc of 4 x 128 MB: ~50 ms)
        const size_t per = (size_t)T * p->F;
        // (both chunkings: a call that is one batch stage, and one with a one-workgroup-per-spectrogram stage -- larger chunks)
        const int bc0 = host_chunk(per, B, cu_count(p->device), false), bc1 = host_chunk(per, B, cu_count(p->device), true);
        if ((rc = p->pipe.ensure((size_t)bc0 * per * sizeof(float2), bc0 < B ? 2 : 1))) return rc;
        if ((rc = p->pipe.ensure((size_t)bc1 * per * sizeof(float2), bc1 < B ? 2 : 1))) return rc;
    }
    if ((rc = p->resid_rows.ensure((size_t)B * T * 2 * sizeof(double)))) return rc;
    if ((rc = p->resid_out.ensure((size_t)B * 2 * sizeof(double)))) return rc;
    if (!p->fp64) {
        if (p->sysb) {
            const size_t n_part = p->sysb->io_partials(p->sys, T);
            if ((rc = p->row_sums.ensure((size_t)B * (n_part > (size_t)T ? n_part : (size_t)T) * sizeof(double)))) return rc;
            hipError_t e = p->sysb->reserve(p->sys, B, T, max_iters);
            if (e != hipSuccess) return fail(LWS_ERR_NOMEM, "systolic scratch: %s", hipGetErrorString(e));
        }
    }
    return LWS_OK;
}

int lws_residual_dev(lws_plan *p, const void *S_dev, int B, int T, double *out, void *stream) {
    if (!p || !S_dev || !out) return fail(LWS_ERR_INVALID, "null argument");
    if (B <= 0 || T < 1) return fail(LWS_ERR_INVALID, "need B >= 1 and T >= 1");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc;
    if ((rc = order_after_plan_work(p, s))) return rc;      // (the plan's scratch: behind whatever an earlier *_dev call enqueued)
    if ((rc = p->resid_rows.ensure((size_t)B * T * 2 * sizeof(double)))) return rc;
    if ((rc = p->resid_out.ensure((size_t)B * 2 * sizeof(double)))) return rc;
    if (p->fp64) {
        if ((rc = ensure_scratch<double>(p, B, T, 1))) return rc;
        const double2 *in = static_cast<const double2 *>(S_dev);
        HIP_TRY((lws::launch_prep<double, double2>(in, static_cast<double2 *>(p->state.p),
                                                   static_cast<double *>(p->amp.p),
                                                   static_cast<double *>(p->row_sums.p),
                                                   static_cast<double *>(p->mean_amp.p), B, T, p->F,
                                                   p->L, p->Q, s)));
        HIP_TRY(lws::launch_residual<double>(static_cast<const double2 *>(p->state.p), wset<double>(p, 0),
                                             static_cast<double *>(p->resid_rows.p),
                                             static_cast<double *>(p->resid_out.p), B, T, p->F, p->L,
                                             p->Q, p->Qp, s));
    } else {
        if ((rc = ensure_scratch<float>(p, B, T, 1))) return rc;
        const float2 *in = static_cast<const float2 *>(S_dev);
        HIP_TRY((lws::launch_prep<float, float2>(in, static_cast<float2 *>(p->state.p),
                                                 static_cast<float *>(p->amp.p),
                                                 static_cast<double *>(p->row_sums.p),
                                                 static_cast<double *>(p->mean_amp.p), B, T, p->F, p->L,
                                                 p->Q, s)));
        HIP_TRY(lws::launch_residual<float>(static_cast<const float2 *>(p->state.p), wset<float>(p, 0),
                                            static_cast<double *>(p->resid_rows.p),
                                            static_cast<double *>(p->resid_out.p), B, T, p->F, p->L,
                                            p->Q, p->Qp, s));
    }
    HIP_TRY(hipMemcpyAsync(out, p->resid_out.p, (size_t)B * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LWS_OK;
}

namespace {
// sum of the [B][2] per-spectrogram pairs, in a fixed order (one block, tree over 256 partial sums)
__global__ void __launch_bounds__(256) k_sum_pairs(const double *pairs, double *out, int B) {
    __shared__ double red[2][256];
    double a = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) { a += pairs[2 * b]; c += pairs[2 * b + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = c;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) { red[0][threadIdx.x] += red[0][threadIdx.x + s2]; red[1][threadIdx.x] += red[1][threadIdx.x + s2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = red[0][0]; out[1] = red[1][0]; }
}
// ncclAllReduce of librccl, resolved at first use (the library does not link against RCCL)
using nccl_allreduce_fn = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
nccl_allreduce_fn rccl_allreduce() {
    static std::atomic<nccl_allreduce_fn> fn{nullptr};
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                if (void *sym = dlsym(h, "ncclAllReduce")) { fn.store(reinterpret_cast<nccl_allreduce_fn>(sym)); return; }
            }
        }
    });
    return fn.load();
}
}  // namespace

int lws_weights_structure(const double *W, int Q, int Qp, int L, int *period, int *step) {
    int P = 0, sg = 0;
    if (!W || !period || !step || Q < 2 || Qp < 1 || L < 0) return 0;
    if (!lws::weights_twiddle(W, Q, Qp, L, 4096, &P, &sg)) return 0;
    *period = P; *step = sg;
    return 1;
}

int lws_residual_allreduce_dev(lws_plan *p, const void *S_dev, int B, int T, void *rccl_comm, double *out, void *stream) {
    if (!p || !S_dev || !out) return fail(LWS_ERR_INVALID, "null argument");
    if (B <= 0 || T < 1) return fail(LWS_ERR_INVALID, "need B >= 1 and T >= 1");
    nccl_allreduce_fn allreduce = nullptr;
    if (rccl_comm && !(allreduce = rccl_allreduce())) return fail(LWS_ERR_UNSUPPORTED, "librccl.so.1 (ncclAllReduce) could not be loaded");
    std::vector<double> per((size_t)2 * B);
    int rc = lws_residual_dev(p, S_dev, B, T, per.data(), stream);   // (leaves the [B][2] pairs in p->resid_out on the device)
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((rc = p->resid_sum.ensure(2 * sizeof(double)))) return rc;
    double *sum = static_cast<double *>(p->resid_sum.p);
    hipLaunchKernelGGL(k_sum_pairs, dim3(1), dim3(256), 0, s, static_cast<const double *>(p->resid_out.p), sum, B);
    HIP_TRY(hipGetLastError());
    if (rccl_comm) {
        const int nrc = allreduce(sum, sum, 2, /* ncclFloat64 */ 8, /* ncclSum */ 0, rccl_comm, s);
        if (nrc != 0) return fail(LWS_ERR_HIP, "ncclAllReduce failed (ncclResult_t %d)", nrc);
    }
    HIP_TRY(hipMemcpyAsync(out, sum, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LWS_OK;
}

int lws_residual(lws_plan *p, const double *S, int B, int T, double *out) {
    if (!p || !S || !out) return fail(LWS_ERR_INVALID, "null argument");
    if (B <= 0 || T < 1) return fail(LWS_ERR_INVALID, "need B >= 1 and T >= 1");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = nullptr;
    const size_t count = (size_t)B * T * p->F;
    int rc;
    if ((rc = order_after_plan_work(p, s))) return rc;
    if ((rc = p->stage.ensure(count * sizeof(double2)))) return rc;
    if ((rc = p->resid_rows.ensure((size_t)B * T * 2 * sizeof(double)))) return rc;
    if ((rc = p->resid_out.ensure((size_t)B * 2 * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpyAsync(p->stage.p, S, count * sizeof(double2), hipMemcpyHostToDevice, s));
    const double2 *in = static_cast<const double2 *>(p->stage.p);
    if (p->fp64) {
        if ((rc = ensure_scratch<double>(p, B, T, 1))) return rc;
        HIP_TRY((lws::launch_prep<double, double2>(in, static_cast<double2 *>(p->state.p), static_cast<double *>(p->amp.p),
                                                   static_cast<double *>(p->row_sums.p), static_cast<double *>(p->mean_amp.p),
                                                   B, T, p->F, p->L, p->Q, s)));
        HIP_TRY(lws::launch_residual<double>(static_cast<const double2 *>(p->state.p), wset<double>(p, 0),
                                             static_cast<double *>(p->resid_rows.p), static_cast<double *>(p->resid_out.p), B, T,
                                             p->F, p->L, p->Q, p->Qp, s));
    } else {
        if ((rc = ensure_scratch<float>(p, B, T, 1))) return rc;
        HIP_TRY((lws::launch_prep<float, double2>(in, static_cast<float2 *>(p->state.p), static_cast<float *>(p->amp.p),
                                                  static_cast<double *>(p->row_sums.p), static_cast<double *>(p->mean_amp.p),
                                                  B, T, p->F, p->L, p->Q, s)));
        HIP_TRY(lws::launch_residual<float>(static_cast<const float2 *>(p->state.p), wset<float>(p, 0),
                                            static_cast<double *>(p->resid_rows.p), static_cast<double *>(p->resid_out.p), B, T,
                                            p->F, p->L, p->Q, p->Qp, s));
    }
    HIP_TRY(hipMemcpyAsync(out, p->resid_out.p, (size_t)B * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LWS_OK;
}

int lws_last_kernel_time(lws_plan *p, float *ms, int *launches) {
    if (!p) return fail(LWS_ERR_INVALID, "null plan");
    if (p->timing_pending) {
        HIP_TRY(hipEventSynchronize(p->ev1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, p->ev0, p->ev1));
        p->last_ms = t;
        p->timing_pending = false;
        int rc = check_systolic_flag(p);
        if (rc) return rc;
    }
    if (ms) *ms = p->last_ms;
    if (launches) *launches = p->last_launches;
    return LWS_OK;
}

const char *lws_last_kernel_name(lws_plan *p) { return p ? p->last_name : "none"; }

const char *lws_generic_stage(lws_plan *p) { return p ? p->generic_stage : ""; }

// Test hook, not part of include/lws_hip.h: the fp64 systolic engine's scratch layout for frames of F bins (tests/test_sys64_model.py
// checks that no prefetch reads past the rows it allocates).  out = {rows, highest row read, highest row written, gap}; 0 if the shape
// is not one the engine takes.
int lws_debug_sys64_layout(int F, int T, int Q, long *out) { return out && lws::sys64_layout(F, T, Q, out) ? 1 : 0; }

// Test hook, not part of include/lws_hip.h: ONE stage on extended buffers the caller supplies -- the reference's ExtSr/ExtSi and AmpSpec
// (lws.pyx:235-240) as it would hand them to a kernel of lwslib.h, [B][T + 2(Q-1)][F + 2L] on the host, in the plan's arithmetic type
// (complex64 + float32, or complex128 + float64) -- with the thresholds taken as they are (not scaled by mean|S|).  State and target
// magnitudes are independent here, which no public entry point allows: a frame whose targets are zero is never updated, so a test can
// freeze the first m0 frames at values of its choice and compare what the PRODUCTION kernel makes of the frames after them with the
// CPU restatement of the reference on the same buffers (tests/test_gpu_teacher.py).  stage: 0 batch, 1 no-future, 2 online.  The state is updated in place.
int lws_debug_stage_ext(lws_plan *p, int stage, int wsel, void *state_ext, const void *amp_ext, int B, int T, const double *thresholds, int iters,
                        int LA, double qdiv) {
    int rc = check_common(p, B, T, thresholds, iters);
    if (rc) return rc;
    if (!state_ext || !amp_ext || stage < 0 || stage > 2 || B < 1 || iters < 1) return fail(LWS_ERR_INVALID, "lws_debug_stage_ext: bad arguments");
    StageSpec st{stage == 0 ? lws::MODE_BATCH : (stage == 1 ? lws::MODE_NOFUTURE : lws::MODE_ONLINE), wsel, thresholds, iters, LA, qdiv};
    if ((rc = need_weights(p, &st, 1))) return rc;
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = nullptr;
    if ((rc = order_after_plan_work(p, s))) return rc;
    rc = p->fp64 ? ensure_scratch<double>(p, B, T, iters) : ensure_scratch<float>(p, B, T, iters);
    if (rc) return rc;
    const size_t n = (size_t)B * (T + 2 * (p->Q - 1)) * (p->F + 2 * p->L), rs = p->fp64 ? sizeof(double) : sizeof(float);
    HIP_TRY(hipMemcpy(p->state.p, state_ext, n * 2 * rs, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->amp.p, amp_ext, n * rs, hipMemcpyHostToDevice));
    std::vector<double> ones((size_t)B, 1.0);
    HIP_TRY(hipMemcpy(p->mean_amp.p, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    rc = p->fp64 ? run_stage<double>(p, st.mode, wsel, B, T, thresholds, iters, LA, qdiv, s) : run_stage<float>(p, st.mode, wsel, B, T, thresholds, iters, LA, qdiv, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(state_ext, p->state.p, n * 2 * rs, hipMemcpyDeviceToHost));
    return LWS_OK;
}

}  // extern "C"
