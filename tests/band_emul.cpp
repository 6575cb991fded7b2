// band_emul.cpp -- the band engine's schedule (lws_amd/csrc/lws_band_core.h, the very text the kernel is compiled from) stepped
// through on the CPU, lane by lane: TEST INFRASTRUCTURE (tests/test_band_model.py builds it with g++ and compares it with the
// oracle).  What the GPU does concurrently -- the lanes of a wave, the waves of a workgroup between two barriers -- runs here
// one after the other; that is the same thing as long as no step reads a ring row written in the same step, which
// Lane::row_at asserts (age >= 2 at the time of the next step, for which the read is issued).
#include "../lws_amd/csrc/lws_band_host.h"

#include <complex>
#include <cstdio>
#include <cstring>
#include <memory>

using namespace lws::band;

namespace {
template <typename real> struct Cx { real x, y; };

template <typename real, int LT, int QT>
int run(const double *S, double *out, int T, int F, const double *W, int Q, int L, int Pt, int s, const double *thr, int n_thr, int NS,
        int SKW, int nls) {
    using C = Cx<real>;
    if (Q > QT || L > LT || SKW < LT + 2 || (nls & (nls - 1)) || F < 2 * LT + 7) return 2;
    const Geom g = geometry(F, T, Q, LT, SKW, nls, Pt);
    const int Tp = T + 2 * (Q - 1);
    std::vector<double> wtd, twd;
    tables(W, Q, L, LT, Pt, s, wtd, twd);
    std::vector<C> wt(wtd.size() / 2), tw(twd.size() / 2);
    for (size_t i = 0; i < wt.size(); ++i) { wt[i].x = (real)wtd[2 * i]; wt[i].y = (real)wtd[2 * i + 1]; }
    for (size_t i = 0; i < tw.size(); ++i) { tw[i].x = (real)twd[2 * i]; tw[i].y = (real)twd[2 * i + 1]; }
    // the skewed state: frame me (clamped at both ends, lwslib.cpp:15-40), bin b at row SKW (me % nls) + P (me / nls) + b + LT
    std::vector<C> G((size_t)g.rows * nls, C{0, 0});
    std::vector<real> A((size_t)g.rows * nls, (real)0);
    for (int me = 0; me < Tp; ++me) {
        const int src = std::min(std::max(me - (Q - 1), 0), T - 1);
        const long base = (long)SKW * (me % nls) + (long)g.P * (me / nls) + LT;
        for (int b = 0; b < F + LT; ++b) {
            const int bb = b < F ? b : 2 * (F - 1) - b;
            C v;
            v.x = (real)S[2 * ((size_t)src * F + bb)];
            v.y = (real)(b < F ? S[2 * ((size_t)src * F + bb) + 1] : -S[2 * ((size_t)src * F + bb) + 1]);
            G[(size_t)(base + b) * nls + me % nls] = v;
            A[(size_t)(base + b) * nls + me % nls] = (real)std::hypot(S[2 * ((size_t)src * F + bb)], S[2 * ((size_t)src * F + bb) + 1]);
        }
    }
    std::vector<C> rings((size_t)NS * g.R * nls);
    for (int i0 = 0; i0 < n_thr; i0 += NS) {
        const int ns = std::min(NS, n_thr - i0);
        std::fill(rings.begin(), rings.end(), C{0, 0});
        using L0 = Lane<real, C, LT, QT, true>;
        using L1 = Lane<real, C, LT, QT, false>;
        std::vector<Env<real, C>> env(ns);
        std::vector<std::unique_ptr<L0>> first;
        std::vector<std::vector<std::unique_ptr<L1>>> rest(ns);
        for (int sl = 0; sl < ns; ++sl) {
            Env<real, C> &e = env[sl];
            e.g = g;
            e.ring_own = rings.data() + (size_t)sl * g.R * nls;
            e.ring_prev = rings.data() + (size_t)(sl > 0 ? sl - 1 : 0) * g.R * nls;
            e.tw = tw.data(); e.wt = wt.data();
            e.G = G.data(); e.A = A.data();
            e.mail = nullptr; e.mail_nh = 0; e.mail_h = 0;
            e.thr = (real)thr[i0 + sl];
            e.last = sl == ns - 1;
        }
        for (int ln = 0; ln < nls; ++ln) first.emplace_back(new L0(env[0], ln, 0));
        for (int sl = 1; sl < ns; ++sl)
            for (int ln = 0; ln < nls; ++ln) rest[sl].emplace_back(new L1(env[sl], ln, sl));
        const int t_end = g.U + g.LAG * (ns - 1);
        for (int t = 0; t < t_end; ++t)
            for (int sl = 0; sl < ns; ++sl) {
                const int u = t - g.LAG * sl;
                if (u < 0 || u >= g.U) continue;
                const int ph = u % SKW;
                for (int ln = 0; ln < nls; ++ln) {
                    if (sl == 0) {
                        if (u == 0) first[ln]->prologue();
                        if (u & 1) first[ln]->template step<1>(u, ph); else first[ln]->template step<0>(u, ph);
                    } else {
                        if (u == 0) rest[sl][ln]->prologue();
                        if (u & 1) rest[sl][ln]->template step<1>(u, ph); else rest[sl][ln]->template step<0>(u, ph);
                    }
                }
            }
    }
    for (int m = 0; m < T; ++m) {
        const int me = m + Q - 1;
        const long base = (long)SKW * (me % nls) + (long)g.P * (me / nls) + LT;
        for (int b = 0; b < F; ++b) {
            const C v = G[(size_t)(base + b) * nls + me % nls];
            out[2 * ((size_t)m * F + b)] = v.x;
            out[2 * ((size_t)m * F + b) + 1] = v.y;
        }
        // the images above Nyquist the pass left in the skewed state are those of the final values
        for (int j = 1; j <= LT; ++j) {
            const C im = G[(size_t)(base + F - 1 + j) * nls + me % nls], sv = G[(size_t)(base + F - 1 - j) * nls + me % nls];
            if (im.x != sv.x || im.y != -sv.y) return 3;
        }
    }
    return 0;
}
// the exact build with helper waves (lws_band_core.h: Split): a slot = a main wave and NH helpers a step ahead of it
template <typename real, int LT, int QT>
int run_helpers(const double *S, double *out, int T, int F, const double *W, int Q, int L, int Pt, int s, const double *thr, int n_thr, int NS,
                int SKW, int nls) {
    using C = Cx<real>;
    using SP = Split<QT>;
    constexpr int NH = SP::NH;
    if (Q != QT || NH < 1 || L > LT || SKW < LT + 2 || (nls & (nls - 1)) || F < 2 * LT + 7) return 2;
    const Geom g = geometry(F, T, Q, LT, SKW, nls, Pt, NH);
    const int Tp = T + 2 * (Q - 1);
    std::vector<double> wtd, twd;
    tables(W, Q, L, LT, Pt, s, wtd, twd);
    std::vector<C> wt(wtd.size() / 2), tw(twd.size() / 2);
    for (size_t i = 0; i < wt.size(); ++i) { wt[i].x = (real)wtd[2 * i]; wt[i].y = (real)wtd[2 * i + 1]; }
    for (size_t i = 0; i < tw.size(); ++i) { tw[i].x = (real)twd[2 * i]; tw[i].y = (real)twd[2 * i + 1]; }
    std::vector<C> G((size_t)g.rows * nls, C{0, 0});
    std::vector<real> A((size_t)g.rows * nls, (real)0);
    for (int me = 0; me < Tp; ++me) {
        const int src = std::min(std::max(me - (Q - 1), 0), T - 1);
        const long base = (long)SKW * (me % nls) + (long)g.P * (me / nls) + LT;
        for (int b = 0; b < F + LT; ++b) {
            const int bb = b < F ? b : 2 * (F - 1) - b;
            C v;
            v.x = (real)S[2 * ((size_t)src * F + bb)];
            v.y = (real)(b < F ? S[2 * ((size_t)src * F + bb) + 1] : -S[2 * ((size_t)src * F + bb) + 1]);
            G[(size_t)(base + b) * nls + me % nls] = v;
            A[(size_t)(base + b) * nls + me % nls] = (real)std::hypot(S[2 * ((size_t)src * F + bb)], S[2 * ((size_t)src * F + bb) + 1]);
        }
    }
    using M0 = Lane<real, C, LT, QT, true, true, false, SP::lo(0), SP::hi(0), NH>;
    using M1 = Lane<real, C, LT, QT, false, true, false, SP::lo(0), SP::hi(0), NH>;
    using H0a = Lane<real, C, LT, QT, true, true, true, SP::lo(1), SP::hi(1), 0>;
    using H1a = Lane<real, C, LT, QT, false, true, true, SP::lo(1), SP::hi(1), 0>;
    using H0b = Lane<real, C, LT, QT, true, true, true, SP::lo(NH >= 2 ? 2 : 1), SP::hi(NH >= 2 ? 2 : 1), 0>;
    using H1b = Lane<real, C, LT, QT, false, true, true, SP::lo(NH >= 2 ? 2 : 1), SP::hi(NH >= 2 ? 2 : 1), 0>;
    using H0c = Lane<real, C, LT, QT, true, true, true, SP::lo(NH >= 3 ? 3 : 1), SP::hi(NH >= 3 ? 3 : 1), 0>;
    using H1c = Lane<real, C, LT, QT, false, true, true, SP::lo(NH >= 3 ? 3 : 1), SP::hi(NH >= 3 ? 3 : 1), 0>;
    std::vector<C> rings((size_t)NS * g.R * nls), mail((size_t)NS * 2 * NH * nls * 2);
    for (int i0 = 0; i0 < n_thr; i0 += NS) {
        const int ns = std::min(NS, n_thr - i0);
        std::fill(rings.begin(), rings.end(), C{0, 0});
        std::fill(mail.begin(), mail.end(), C{0, 0});
        std::vector<Env<real, C>> env((size_t)ns * (1 + NH));
        for (int sl = 0; sl < ns; ++sl)
            for (int h = 0; h <= NH; ++h) {
                Env<real, C> &e = env[(size_t)sl * (1 + NH) + h];
                e.g = g;
                e.ring_own = rings.data() + (size_t)sl * g.R * nls;
                e.ring_prev = rings.data() + (size_t)(sl > 0 ? sl - 1 : 0) * g.R * nls;
                e.tw = tw.data(); e.wt = wt.data();
                e.G = G.data(); e.A = A.data();
                e.mail = mail.data() + (size_t)sl * 2 * NH * nls * 2;
                e.mail_nh = NH; e.mail_h = h > 0 ? h - 1 : 0;
                e.thr = (real)thr[i0 + sl];
                e.last = sl == ns - 1;
            }
        std::vector<std::unique_ptr<M0>> m0; std::vector<std::unique_ptr<H0a>> h0a; std::vector<std::unique_ptr<H0b>> h0b; std::vector<std::unique_ptr<H0c>> h0c;
        std::vector<std::vector<std::unique_ptr<M1>>> m1(ns); std::vector<std::vector<std::unique_ptr<H1a>>> h1a(ns);
        std::vector<std::vector<std::unique_ptr<H1b>>> h1b(ns); std::vector<std::vector<std::unique_ptr<H1c>>> h1c(ns);
        for (int ln = 0; ln < nls; ++ln) {
            m0.emplace_back(new M0(env[0], ln, 0)); h0a.emplace_back(new H0a(env[1], ln, 0));
            if (NH >= 2) h0b.emplace_back(new H0b(env[2], ln, 0));
            if (NH >= 3) h0c.emplace_back(new H0c(env[3], ln, 0));
        }
        for (int sl = 1; sl < ns; ++sl)
            for (int ln = 0; ln < nls; ++ln) {
                m1[sl].emplace_back(new M1(env[(size_t)sl * (1 + NH)], ln, sl)); h1a[sl].emplace_back(new H1a(env[(size_t)sl * (1 + NH) + 1], ln, sl));
                if (NH >= 2) h1b[sl].emplace_back(new H1b(env[(size_t)sl * (1 + NH) + 2], ln, sl));
                if (NH >= 3) h1c[sl].emplace_back(new H1c(env[(size_t)sl * (1 + NH) + 3], ln, sl));
            }
        auto go = [&](auto &lane, int u) {
            if (u < 0 || u >= g.U) return;
            if (u == 0) lane.prologue();
            if (u & 1) lane.template step<1>(u, u % SKW); else lane.template step<0>(u, u % SKW);
        };
        const int t_end = g.U + g.LAG * (ns - 1) + 2;
        for (int t = 0; t < t_end; ++t)
            for (int sl = 0; sl < ns; ++sl) {
                const int ua = t - g.LAG * sl - 2, ub = ua + 1;       // the main wave's frame-time, the helpers'
                for (int ln = 0; ln < nls; ++ln) {
                    if (sl == 0) {
                        go(*m0[ln], ua); go(*h0a[ln], ub);
                        if (NH >= 2) go(*h0b[ln], ub);
                        if (NH >= 3) go(*h0c[ln], ub);
                    } else {
                        go(*m1[sl][ln], ua); go(*h1a[sl][ln], ub);
                        if (NH >= 2) go(*h1b[sl][ln], ub);
                        if (NH >= 3) go(*h1c[sl][ln], ub);
                    }
                }
            }
    }
    for (int m = 0; m < T; ++m) {
        const int me = m + Q - 1;
        const long base = (long)SKW * (me % nls) + (long)g.P * (me / nls) + LT;
        for (int b = 0; b < F; ++b) {
            const C v = G[(size_t)(base + b) * nls + me % nls];
            out[2 * ((size_t)m * F + b)] = v.x;
            out[2 * ((size_t)m * F + b) + 1] = v.y;
        }
    }
    return 0;
}
}  // namespace

extern "C" int band_emul_helpers(const double *S, double *out, int T, int F, const double *W, int Q, int L, int Pt, int s, const double *thr,
                                 int n_thr, int NS, int SKW, int nls, int LT, int QT, int fp32) {
#define BAND_HCASE(LT_, QT_)                                                                                               \
    if (LT == LT_ && QT == QT_)                                                                                            \
        return fp32 ? run_helpers<float, LT_, QT_>(S, out, T, F, W, Q, L, Pt, s, thr, n_thr, NS, SKW, nls)                  \
                    : run_helpers<double, LT_, QT_>(S, out, T, F, W, Q, L, Pt, s, thr, n_thr, NS, SKW, nls);
    BAND_HCASE(5, 8)
    BAND_HCASE(5, 16)
    BAND_HCASE(10, 4)
    BAND_HCASE(8, 4)
    return 1;
}

// S, out: [T][F] complex128; W: row 0 of the plan's tensor, [Q][L+1] complex128; thr: thresholds already scaled by mean|S|
extern "C" int band_emul(const double *S, double *out, int T, int F, const double *W, int Q, int L, int Pt, int s, const double *thr,
                         int n_thr, int NS, int SKW, int nls, int LT, int QT, int fp32) {
#define BAND_CASE(LT_, QT_)                                                                                                \
    if (LT == LT_ && QT == QT_)                                                                                            \
        return fp32 ? run<float, LT_, QT_>(S, out, T, F, W, Q, L, Pt, s, thr, n_thr, NS, SKW, nls)                          \
                    : run<double, LT_, QT_>(S, out, T, F, W, Q, L, Pt, s, thr, n_thr, NS, SKW, nls);
    BAND_CASE(5, 8)
    BAND_CASE(5, 16)
    BAND_CASE(10, 8)
    BAND_CASE(10, 16)
    return 1;
}
extern "C" void band_emul_geometry(int F, int T, int Q, int LT, int SKW, int nls, int Pt, long *out) {
    const Geom g = geometry(F, T, Q, LT, SKW, nls, Pt);
    out[0] = g.P; out[1] = g.gap; out[2] = g.LAG; out[3] = g.R; out[4] = g.nblk; out[5] = g.U; out[6] = g.rows;
}
