// lwslib_compat.hip -- the reference's native interface (include/lwslib_compat.h, == lwslib/lwslib.h:6-26) on the
// order-exact generic HIP engine, in fp64.  One call = upload the frames the call can touch, run one sweep (or the
// whole online driver), download the updated frames.  Compatibility shims: see the header for what they are for.
#include "../../include/lwslib_compat.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "lws_common.h"

namespace {

thread_local std::string g_compat_err;

struct Dev {
    void *p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 8) == hipSuccess; }
};

bool fail(const char *what, hipError_t e) {
    g_compat_err = std::string(what) + ": " + hipGetErrorString(e);
    fprintf(stderr, "lwslib_compat: %s\n", g_compat_err.c_str());
    return false;
}

// split fp64 weights + int flags -> device double2 (zero where the flag is off) + uint8 flags
bool upload_weights(const double *wr, const double *wi, const int *flag, size_t n, Dev &dw, Dev &df) {
    std::vector<double2> w(n);
    std::vector<uint8_t> f(n);
    for (size_t i = 0; i < n; ++i) {
        f[i] = flag[i] ? 1 : 0;
        w[i].x = flag[i] ? wr[i] : 0.0;
        w[i].y = flag[i] ? wi[i] : 0.0;
    }
    if (!dw.alloc(n * sizeof(double2)) || !df.alloc(n)) return fail("hipMalloc", hipErrorOutOfMemory);
    hipError_t e = hipMemcpy(dw.p, w.data(), n * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(df.p, f.data(), n, hipMemcpyHostToDevice);
    return e == hipSuccess ? true : fail("hipMemcpy(weights)", e);
}

// frames (relative to the pointers) a sweep over M frames with M0 usable right frames can read: the last one
int rows_touched(int M, int M0, int Q) {
    int last = 0;
    for (int j = 0; j < M; ++j) {
        int ts = M0 - j;
        if (ts > Q) ts = Q;
        if (ts < 1) ts = 1;
        const int r = Q - 1 + j + (ts - 1);
        if (r > last) last = r;
    }
    return last + 1;
}

// One sweep.  mode: lws::Mode.  For MODE_BATCH M0 is ignored (all right frames), MODE_NOFUTURE(_Q4_COMPAT): none.
void run_sweep(int mode, double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *Amp, int F, int M, int M0,
               int L, int Q, int Qp, double threshold, int update, double qdiv) {
    if (M <= 0) return;
    const int Np = F + 2 * L;
    const int Tp = M + 2 * (Q - 1);
    int rows = Tp;
    if (mode == lws::MODE_ASYM) rows = rows_touched(M, M0, Q);
    if (mode == lws::MODE_NOFUTURE || mode == lws::MODE_NOFUTURE_Q4_COMPAT) rows = M + Q - 1;
    if (rows > Tp) rows = Tp;
    const size_t n_rows = (size_t)rows * Np, n_all = (size_t)Tp * Np;
    std::vector<double2> st(n_all);
    std::vector<double> am(n_all, 0.0);
    for (size_t i = 0; i < n_rows; ++i) { st[i].x = Sr[i]; st[i].y = Si[i]; am[i] = Amp[i]; }
    for (size_t i = n_rows; i < n_all; ++i) { st[i].x = 0; st[i].y = 0; }
    Dev dst, dam, dth, dw, df;
    if (!dst.alloc(n_all * sizeof(double2)) || !dam.alloc(n_all * sizeof(double)) || !dth.alloc(sizeof(double))) {
        fail("hipMalloc", hipErrorOutOfMemory);
        return;
    }
    if (!upload_weights(wr, wi, w_flag, (size_t)Qp * Q * (L + 1), dw, df)) return;
    hipError_t e = hipMemcpy(dst.p, st.data(), n_all * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dam.p, am.data(), n_all * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dth.p, &threshold, sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("hipMemcpy(in)", e); return; }
    lws::GenericArgs<double> a;
    a.state = static_cast<double2 *>(dst.p);
    a.amp = static_cast<const double *>(dam.p);
    a.thr = static_cast<const double *>(dth.p);
    for (int i = 0; i < 3; ++i) { a.w[i].w = static_cast<const double2 *>(dw.p); a.w[i].flag = static_cast<const uint8_t *>(df.p); }
    a.wsel = 0;
    a.F = F; a.T = M; a.L = L; a.Q = Q; a.Qp = Qp;
    a.n_thr = 1; a.LA = 0; a.M0 = M0; a.update = update; a.qdiv = qdiv; a.mode = mode; a.group = 1;
    e = lws::launch_generic<double>(a, 1, nullptr);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst.p, n_all * sizeof(double2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { fail("generic sweep", e); return; }
    // only frames Q-1 .. M+Q-2 (all their columns, images included) can have changed
    for (size_t i = (size_t)(Q - 1) * Np; i < (size_t)(M + Q - 1) * Np; ++i) { Sr[i] = st[i].x; Si[i] = st[i].y; }
}

}  // namespace

const char *lwslib_compat_last_error(void) { return g_compat_err.c_str(); }

// ---- helpers (host loops, as in the reference) ----
void ExtendSpec(double *ExtSr, double *ExtSi, double *InSr, double *InSi, int Nreal, int M, int L, int Q) {
    const int Np = Nreal + 2 * L, nyq = Nreal + L - 1;
    for (int me = 0; me < M + 2 * (Q - 1); ++me) {
        int src = me - (Q - 1);
        if (src < 0) src = 0;
        if (src > M - 1) src = M - 1;
        double *er = ExtSr + (size_t)me * Np, *ei = ExtSi + (size_t)me * Np;
        for (int c = 0; c < Nreal; ++c) { er[L + c] = InSr[(size_t)src * Nreal + c]; ei[L + c] = InSi[(size_t)src * Nreal + c]; }
        for (int j = 1; j <= L; ++j) {
            er[L - j] = er[L + j]; ei[L - j] = -ei[L + j];
            er[nyq + j] = er[nyq - j]; ei[nyq + j] = -ei[nyq - j];
        }
    }
}
void CopySpec(double *ExtSr, double *ExtSi, double *InSr, double *InSi, int Nreal, int M, int L, int Q) {
    const int Np = Nreal + 2 * L;
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < Nreal; ++c) {
            InSr[(size_t)m * Nreal + c] = ExtSr[(size_t)(m + Q - 1) * Np + L + c];
            InSi[(size_t)m * Nreal + c] = ExtSi[(size_t)(m + Q - 1) * Np + L + c];
        }
}
void ComputeAmpSpec(double *Sr, double *Si, double *AmpSpec, int size) {
    for (int i = 0; i < size; ++i) AmpSpec[i] = sqrt(Sr[i] * Sr[i] + Si[i] * Si[i]);
}

// ---- batch ----
void LWSQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, double threshold) {
    run_sweep(lws::MODE_BATCH, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, 2, 2, threshold, 2, 2.0);
}
void LWSQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, double threshold) {
    run_sweep(lws::MODE_BATCH, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, 4, 4, threshold, 2, 4.0);
}
void LWSanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, int Q, double threshold) {
    run_sweep(lws::MODE_BATCH, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, Q, Q, threshold, 2, (double)Q);
}
void LWSfractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, int Q, double threshold) {
    run_sweep(lws::MODE_BATCH, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, Q, 2 * (Nreal - 1), threshold, 2, (double)Q);
}

// ---- no future ----
void NoFuture_LWSQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, double threshold) {
    run_sweep(lws::MODE_NOFUTURE, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, 2, 2, threshold, 2, 2.0);
}
void NoFuture_LWSQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, double threshold) {
    run_sweep(lws::MODE_NOFUTURE_Q4_COMPAT, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, 4, 4, threshold, 2, 4.0);
}
void NoFuture_LWSanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, int Q, double threshold) {
    run_sweep(lws::MODE_NOFUTURE, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, Q, Q, threshold, 2, (double)Q);
}
void NoFuture_LWSfractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int L, int Q, double threshold) {
    run_sweep(lws::MODE_NOFUTURE, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, 0, L, Q, 2 * (Nreal - 1), threshold, 2, (double)Q);
}

// ---- asymmetric ----
void Asym_UpdatePhaseQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int M0, int L, double threshold, int update) {
    run_sweep(lws::MODE_ASYM, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, M0, L, 2, 2, threshold, update, 2.0);
}
void Asym_UpdatePhaseQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int M0, int L, double threshold, int update) {
    run_sweep(lws::MODE_ASYM, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, M0, L, 4, 4, threshold, update, 4.0);
}
void Asym_UpdatePhaseanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int M0, int L, int Q, double threshold, int update) {
    run_sweep(lws::MODE_ASYM, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, M0, L, Q, Q, threshold, update, (double)Q);
}
void Asym_UpdatePhasefractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec, int Nreal, int M, int M0, int L, int Q, double Qfloat, double threshold, int update) {
    run_sweep(lws::MODE_ASYM, Sr, Si, wr, wi, w_flag, AmpSpec, Nreal, M, M0, L, Q, 2 * (Nreal - 1), threshold, update, Qfloat);
}

// ---- online driver: the whole of TF_RTISI_LA in one launch ----
void TF_RTISI_LA(double *Sr, double *Si, double *wr, double *wi, double *wr_asym_init, double *wi_asym_init,
                 double *wr_asym_full, double *wi_asym_full, int *w_flag, int *w_flag_ai, int *w_flag_af,
                 double *AmpSpec, int iter, int LA, int Nreal, int M, int L, int Q, double Qfloat,
                 int use_summarized_weights, double *ThresholdArray, int update) {
    if (M <= 0) return;
    const int F = Nreal, Np = F + 2 * L, Tp = M + 2 * (Q - 1);
    const int Qp = use_summarized_weights ? Q : 2 * (F - 1);
    const size_t n_all = (size_t)Tp * Np, nw = (size_t)Qp * Q * (L + 1);
    std::vector<double2> st(n_all);
    for (size_t i = 0; i < n_all; ++i) { st[i].x = Sr[i]; st[i].y = Si[i]; }
    Dev dst, dam, dth, dw[3], df[3];
    if (!dst.alloc(n_all * sizeof(double2)) || !dam.alloc(n_all * sizeof(double)) || !dth.alloc(sizeof(double) * (iter > 0 ? iter : 1))) {
        fail("hipMalloc", hipErrorOutOfMemory);
        return;
    }
    if (!upload_weights(wr, wi, w_flag, nw, dw[0], df[0]) || !upload_weights(wr_asym_init, wi_asym_init, w_flag_ai, nw, dw[1], df[1]) ||
        !upload_weights(wr_asym_full, wi_asym_full, w_flag_af, nw, dw[2], df[2]))
        return;
    hipError_t e = hipMemcpy(dst.p, st.data(), n_all * sizeof(double2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dam.p, AmpSpec, n_all * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && iter > 0) e = hipMemcpy(dth.p, ThresholdArray, sizeof(double) * iter, hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("hipMemcpy(in)", e); return; }
    lws::GenericArgs<double> a;
    a.state = static_cast<double2 *>(dst.p);
    a.amp = static_cast<const double *>(dam.p);
    a.thr = static_cast<const double *>(dth.p);
    for (int i = 0; i < 3; ++i) { a.w[i].w = static_cast<const double2 *>(dw[i].p); a.w[i].flag = static_cast<const uint8_t *>(df[i].p); }
    a.wsel = 0;
    a.F = F; a.T = M; a.L = L; a.Q = Q; a.Qp = Qp;
    a.n_thr = iter; a.LA = LA; a.M0 = 0; a.update = update;
    a.qdiv = use_summarized_weights ? (double)Q : Qfloat;
    a.mode = lws::MODE_ONLINE; a.group = 1;
    e = lws::launch_generic<double>(a, 1, nullptr);
    if (e == hipSuccess) e = hipMemcpy(st.data(), dst.p, n_all * sizeof(double2), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { fail("online driver", e); return; }
    for (size_t i = 0; i < n_all; ++i) { Sr[i] = st[i].x; Si[i] = st[i].y; }
}
