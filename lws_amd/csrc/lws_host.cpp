// lws_host.cpp -- host-side construction of windows, weights and threshold schedules behind the C ABI, so that callers
// without numpy (C++, the mex gateways) can build a plan: hann, synthwin, create_weights, build_asymmetric_windows,
// get_thresholds of lws.pyx:10-40,160-206 and the window / weight set-up of `class lws` (lws.pyx:384-447).
// Plain fp64 C++; no device code.
#include "../../include/lws_hip.h"

#include <cmath>
#include <vector>

#include "lws_common.h"

namespace {

int ceil_div(int a, int b) { return (a + b - 1) / b; }

// lws.pyx:10-19
void hann_impl(int n, bool symmetric, bool use_offset, double *out) {
    for (int i = 0; i < n; ++i) {
        const double phase = symmetric ? (2.0 * i + 1.0) / (2.0 * n) : (double)(i + (use_offset ? 1 : 0)) / (double)n;
        out[i] = 0.5 * (1.0 - std::cos(2.0 * M_PI * phase));
    }
}

// lws.pyx:22-40; returns false if the normaliser is not strictly positive
bool synthwin_impl(const double *awin, int fsize, int fshift, const double *swin, double *out) {
    const int Q = ceil_div(fsize, fshift);
    if (!swin) swin = awin;
    std::vector<double> overlap(fshift, 0.0);
    for (int q = 0; q < Q; ++q)
        for (int i = 0; i < fshift; ++i) {
            const int t = q * fshift + i;
            if (t < fsize) overlap[i] += awin[t] * swin[t];
        }
    for (int t = 0; t < fsize; ++t)
        if (!(overlap[t % fshift] > 0.0)) return false;
    for (int t = 0; t < fsize; ++t) out[t] = swin[t] / overlap[t % fshift];
    return true;
}

// lws.pyx:160-181.  W: complex128 interleaved, [Qp][Q][L+1]
void create_weights_impl(const double *awin, const double *swin, int T, int fshift, int L, bool summarized, double *W) {
    const int Q = ceil_div(T, fshift);
    const double Qf = (double)T / (double)fshift;
    const int Qp = (T % fshift == 0 && summarized) ? Q : T;
    const int K1 = L + 1;
    std::vector<double> br((size_t)K1 * Q), bi((size_t)K1 * Q);
    for (int l = 0; l <= L; ++l)
        for (int q = 0; q < Q; ++q) {
            const int n = T - q * fshift;
            double sr = 0, si = 0;
            for (int t = 0; t < n; ++t) {
                const double v = awin[t] * swin[q * fshift + t] / T;
                // exp(-2j pi l t / T): reduce l*t modulo T before the trigonometric call
                const double ang = -2.0 * M_PI * (double)(((long long)l * t) % T) / (double)T;
                sr += v * std::cos(ang);
                si += v * std::sin(ang);
            }
            const double a2 = -2.0 * M_PI * (double)l * (double)q / Qf;
            const double c = std::cos(a2), s = std::sin(a2);
            br[(size_t)l * Q + q] = sr * c - si * s;
            bi[(size_t)l * Q + q] = sr * s + si * c;
        }
    br[0] -= 1.0;
    for (int p = 0; p < Qp; ++p)
        for (int q = 0; q < Q; ++q) {
            const double a = 2.0 * M_PI * (double)p * (double)q / Qf;
            const double c = std::cos(a), s = std::sin(a);
            for (int l = 0; l <= L; ++l) {
                const double xr = br[(size_t)l * Q + q], xi = bi[(size_t)l * Q + q];
                double *o = W + 2 * (((size_t)p * Q + q) * K1 + l);
                o[0] = xr * c - xi * s;
                o[1] = xr * s + xi * c;
            }
        }
}

// lws.pyx:184-200
void asym_impl(const double *ws, int T, int fshift, double *win_ai, double *win_af) {
    const int Q = ceil_div(T, fshift);
    for (int t = 0; t < T; ++t) {
        double all = 0, from1 = 0;
        for (int q = 0; q < Q; ++q) {
            const int idx = q * fshift + t;
            if (t < T - q * fshift) {
                all += ws[idx];
                if (q >= 1) from1 += ws[idx];
            }
        }
        win_af[T - 1 - t] = all;
        win_ai[T - 1 - t] = from1;
    }
    if (T % fshift == 2)   // kept verbatim from lws.pyx:198 (the MATLAB binding tests Q == 2 instead)
        for (int t = 0; t < T; ++t) win_ai[t] = ws[t];
}

}  // namespace

extern "C" {

int lws_hann(int n, int symmetric, int use_offset, double *out) {
    if (n < 1 || !out) return lws::set_error(LWS_ERR_INVALID, "hann: n = %d", n);
    hann_impl(n, symmetric != 0, use_offset != 0, out);
    return LWS_OK;
}

int lws_synthwin(const double *awin, int fsize, int fshift, const double *swin, double *out) {
    if (!awin || !out || fsize < 1 || fshift < 1) return lws::set_error(LWS_ERR_INVALID, "synthwin: bad arguments");
    if (!synthwin_impl(awin, fsize, fshift, swin, out))
        return lws::set_error(LWS_ERR_INVALID, "The normalizer is not strictly positive");   // lws.pyx:36
    return LWS_OK;
}

int lws_weights_shape(int fsize, int fshift, int use_summarized_weights, int *Qprime, int *Q) {
    if (fsize < 1 || fshift < 1 || !Qprime || !Q) return lws::set_error(LWS_ERR_INVALID, "weights_shape: bad arguments");
    *Q = ceil_div(fsize, fshift);
    *Qprime = (fsize % fshift == 0 && use_summarized_weights) ? *Q : fsize;
    return LWS_OK;
}

int lws_create_weights(const double *awin, const double *swin, int fsize, int fshift, int L, int use_summarized_weights,
                       double *W) {
    if (!awin || !swin || !W || fsize < 1 || fshift < 1 || L < 0)
        return lws::set_error(LWS_ERR_INVALID, "create_weights: bad arguments");
    create_weights_impl(awin, swin, fsize, fshift, L, use_summarized_weights != 0, W);
    return LWS_OK;
}

int lws_build_asymmetric_windows(const double *awin_swin, int fsize, int fshift, double *win_ai, double *win_af) {
    if (!awin_swin || !win_ai || !win_af || fsize < 1 || fshift < 1)
        return lws::set_error(LWS_ERR_INVALID, "build_asymmetric_windows: bad arguments");
    asym_impl(awin_swin, fsize, fshift, win_ai, win_af);
    return LWS_OK;
}

int lws_get_thresholds(int iterations, double alpha, double beta, double gamma, double *out) {
    if (iterations < 0 || (iterations > 0 && !out)) return lws::set_error(LWS_ERR_INVALID, "get_thresholds: bad arguments");
    for (int i = 0; i < iterations; ++i) out[i] = alpha * std::exp(-beta * std::pow((double)i, gamma));
    return LWS_OK;
}

int lws_plan_create_from_windows(lws_plan **plan, int device, const double *awin_in, const double *swin_in, int fsize,
                                 int fshift, int L, int symmetric_win, unsigned flags, double *awin_out, double *swin_out) {
    if (!plan || fsize < 2 || fshift < 1 || fshift > fsize || L < 0)
        return lws::set_error(LWS_ERR_INVALID, "plan_create_from_windows: bad arguments");
    if (fsize % 2) return lws::set_error(LWS_ERR_INVALID, "Odd ffts not supported.");
    std::vector<double> awin(fsize), swin(fsize), tmp(fsize);
    if (awin_in) {
        for (int i = 0; i < fsize; ++i) awin[i] = awin_in[i];
    } else {   // lws.pyx:386-388: sqrt-Hann made self-dual for this frame shift
        hann_impl(fsize, symmetric_win != 0, false, tmp.data());
        for (int i = 0; i < fsize; ++i) awin[i] = std::sqrt(tmp[i]);
        if (!synthwin_impl(awin.data(), fsize, fshift, nullptr, tmp.data()))
            return lws::set_error(LWS_ERR_INVALID, "The normalizer is not strictly positive");
        for (int i = 0; i < fsize; ++i) awin[i] = std::sqrt(awin[i] * tmp[i]);
    }
    if (!synthwin_impl(awin.data(), fsize, fshift, swin_in, swin.data()))
        return lws::set_error(LWS_ERR_INVALID, "The normalizer is not strictly positive");
    int Qp = 0, Q = 0;
    lws_weights_shape(fsize, fshift, 1, &Qp, &Q);
    const size_t nw = (size_t)Qp * Q * (L + 1) * 2;
    std::vector<double> W(nw), W_ai(nw), W_af(nw), prod(fsize), win_ai(fsize), win_af(fsize);
    for (int i = 0; i < fsize; ++i) prod[i] = awin[i] * swin[i];
    asym_impl(prod.data(), fsize, fshift, win_ai.data(), win_af.data());
    create_weights_impl(awin.data(), swin.data(), fsize, fshift, L, true, W.data());          // lws.pyx:425-431
    create_weights_impl(win_ai.data(), swin.data(), fsize, fshift, L, true, W_ai.data());
    create_weights_impl(win_af.data(), swin.data(), fsize, fshift, L, true, W_af.data());
    if (awin_out) for (int i = 0; i < fsize; ++i) awin_out[i] = awin[i];
    if (swin_out) for (int i = 0; i < fsize; ++i) swin_out[i] = swin[i];
    return lws_plan_create(plan, device, fsize / 2 + 1, L, Q, Qp, W.data(), W_ai.data(), W_af.data(), flags);
}

}  // extern "C"
