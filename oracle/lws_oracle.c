/*
 * lws_oracle.c -- TEST INFRASTRUCTURE ONLY (see lws_oracle.h for the rules and parity status).
 *
 * Restatement, in one canonical complex form, of the twelve per-bin update kernels and the
 * online driver of the reference (lwslib/lwslib.cpp).  The reference's Q2/Q4 kernels are
 * re-associations of its anyQ kernels (they use W[(Q-b)%Q] = +-W[b%Q]); this file implements the
 * anyQ form once and reaches them through it.  Sweep order is the reference's: frames outer,
 * bins inner, every bin overwritten in place as soon as it is computed (Gauss-Seidel).
 *
 * Notation: for a weight w and spectrogram values b, c
 *     pair(w, b, c) = w*b + conj(w)*c
 * which is what the reference spells out as
 *     re += wr*(br+cr) - wi*(bi-ci);   im += wr*(bi+ci) + wi*(br-cr);     (lwslib.cpp:310-311)
 */
#include "lws_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double re, im; } cplx;

/* a += w*b + conj(w)*c, in the grouped form that cancels exactly when c == conj(b)
 * (Hermitian images, real input): this is the arithmetic identity the reference evaluates. */
static inline void acc_pair(cplx *a, double wr, double wi, double br, double bi, double cr, double ci) {
    a->re += wr * (br + cr) - wi * (bi - ci);
    a->im += wr * (bi + ci) + wi * (br - cr);
}
static inline void acc_mul(cplx *a, double wr, double wi, double sr, double si) {
    /* a += w * s */
    a->re += wr * sr - wi * si;
    a->im += wr * si + wi * sr;
}
static inline void acc_mulc(cplx *a, double wr, double wi, double sr, double si) {
    /* a += conj(w) * s */
    a->re += wr * sr + wi * si;
    a->im += wr * si - wi * sr;
}

/* ---- helpers shared by every entry point (lwslib.cpp:15-65) ---- */

void lwso_extend(double *ext_r, double *ext_i, const double *in_r, const double *in_i,
                 int F, int T, int L, int Q) {
    const int Np = F + 2 * L;
    const int Tp = T + 2 * (Q - 1);
    const int nyq = F + L - 1; /* extended column of the Nyquist bin */
    for (int me = 0; me < Tp; ++me) {
        int src = me - (Q - 1); /* frame the extended row is a copy of (edge frames repeat) */
        if (src < 0) src = 0;
        if (src > T - 1) src = T - 1;
        double *er = ext_r + (size_t)me * Np, *ei = ext_i + (size_t)me * Np;
        const double *ir = in_r + (size_t)src * F, *ii = in_i + (size_t)src * F;
        for (int c = 0; c < F; ++c) { er[L + c] = ir[c]; ei[L + c] = ii[c]; }
        for (int j = 1; j <= L; ++j) {
            /* Hermitian images below DC and above Nyquist */
            er[L - j] = er[L + j];     ei[L - j] = -ei[L + j];
            er[nyq + j] = er[nyq - j]; ei[nyq + j] = -ei[nyq - j];
        }
    }
}

void lwso_extract(const double *ext_r, const double *ext_i, double *out_r, double *out_i,
                  int F, int T, int L, int Q) {
    const int Np = F + 2 * L;
    for (int m = 0; m < T; ++m)
        for (int c = 0; c < F; ++c) {
            out_r[(size_t)m * F + c] = ext_r[(size_t)(m + Q - 1) * Np + L + c];
            out_i[(size_t)m * F + c] = ext_i[(size_t)(m + Q - 1) * Np + L + c];
        }
}

void lwso_amplitude(const double *sr, const double *si, double *amp, int count) {
    for (int i = 0; i < count; ++i) amp[i] = sqrt(pow(sr[i], 2.) + pow(si[i], 2.));
}

/* Magnitude re-projection + Hermitian image upkeep, common tail of every kernel
 * (lwslib.cpp:356-368 and the identical blocks of the other eleven kernels). */
static inline void project_and_mirror(double *rowr, double *rowi, int n, cplx a, double target,
                                      int F, int L) {
    const double mag = sqrt(pow(a.re, 2.) + pow(a.im, 2.));
    if (!(mag > 0)) return;
    rowr[n] = a.re * target / mag;
    rowi[n] = a.im * target / mag;
    const int nyq = F + L - 1;
    if (n >= L + 1 && n < 2 * L + 1) {
        rowr[2 * L - n] = rowr[n];
        rowi[2 * L - n] = -rowi[n];
    } else if (n >= F - 1 && n < nyq) {
        rowr[2 * nyq - n] = rowr[n];
        rowi[2 * nyq - n] = -rowi[n];
    }
}

/* ---- the canonical sweep ---- */

static void sweep_canonical(double *sr, double *si, const double *wr, const double *wi,
                            const int *wf, const double *amp, int F, int M, int M0, int L, int Q,
                            int Qp, double threshold, int update, double qdiv) {
    const int Np = F + 2 * L;
    const int K1 = L + 1, RQ = Q * K1;
    for (int m = Q - 1; m < M + Q - 1; ++m) {
        /* how many frames to the right may be used for this frame (lwslib.cpp:1143-1151) */
        int two_sided = M0 + Q - m - 1; /* r < two_sided uses frames m-r and m+r */
        if (two_sided > Q) two_sided = Q;
        int centre = 1;
        if (two_sided < 1) { centre = 0; two_sided = 1; }

        double *cr = sr + (size_t)m * Np, *ci = si + (size_t)m * Np;
        for (int n = L; n < F + L; ++n) {
            const double target = amp[(size_t)m * Np + n];
            if (!(target > threshold)) continue;
            const int bin = n - L;
            const int row = bin % Qp;                /* bin % Q, or the bin itself for general weights */
            const int rowneg = (Qp - row) % Qp;      /* weights of the mirrored bin (periodic) */
            const double *ar = wr + (size_t)row * RQ, *ai = wi + (size_t)row * RQ;
            const int *af = wf + (size_t)row * RQ;
            const double *br = wr + (size_t)rowneg * RQ, *bi = wi + (size_t)rowneg * RQ;
            const int *bf = wf + (size_t)rowneg * RQ;

            cplx a = {0., 0.};
            if (centre) {
                if (update == 1) { a.re += cr[n] / qdiv; a.im += ci[n] / qdiv; }
                for (int k = 1; k <= L; ++k)
                    if (af[k]) acc_pair(&a, ar[k], ai[k], cr[n - k], ci[n - k], cr[n + k], ci[n + k]);
            }
            for (int r = 1; r < Q; ++r) {
                const double *lr = cr - (size_t)r * Np, *li = ci - (size_t)r * Np; /* frame m-r */
                const double *rr = cr + (size_t)r * Np, *ri = ci + (size_t)r * Np; /* frame m+r */
                const int u = r * K1;
                const int both = r < two_sided;
                if (af[u]) {
                    if (both) acc_pair(&a, ar[u], ai[u], lr[n], li[n], rr[n], ri[n]);
                    else acc_mul(&a, ar[u], ai[u], lr[n], li[n]);
                }
                for (int k = 1; k <= L; ++k) {
                    if (af[u + k]) {
                        if (both) acc_pair(&a, ar[u + k], ai[u + k], lr[n - k], li[n - k], rr[n - k], ri[n - k]);
                        else acc_mul(&a, ar[u + k], ai[u + k], lr[n - k], li[n - k]);
                    }
                    if (bf[u + k]) {
                        if (both) acc_pair(&a, br[u + k], bi[u + k], rr[n + k], ri[n + k], lr[n + k], li[n + k]);
                        else acc_mulc(&a, br[u + k], bi[u + k], lr[n + k], li[n + k]);
                    }
                }
            }
            project_and_mirror(cr, ci, n, a, target, F, L);
        }
    }
}

/* NoFuture_LWSQ4 as shipped (lwslib.cpp:538-617): the row base already contains the bin
 * offset and the bin offset is added again, so the neighbours are read at flat offset
 * (m-r)*Np + 2n +- k.  Reproduced on purpose; see DESIGN.md "bug compatibility". */
static void sweep_nofuture_q4_compat(double *sr, double *si, const double *wr, const double *wi,
                                     const int *wf, const double *amp, int F, int M, int L,
                                     double threshold) {
    const int Q = 4, Np = F + 2 * L, K1 = L + 1, RQ = Q * K1;
    for (int m = Q - 1; m < M + Q - 1; ++m) {
        double *cr = sr + (size_t)m * Np, *ci = si + (size_t)m * Np;
        for (int n = L; n < F + L; ++n) {
            const double target = amp[(size_t)m * Np + n];
            if (!(target > threshold)) continue;
            const int bin = n - L, row = bin % Q;
            const double *ar = wr + (size_t)row * RQ, *ai = wi + (size_t)row * RQ;
            const int *af = wf + (size_t)row * RQ;
            cplx a = {0., 0.};
            for (int r = Q - 1; r > 0; --r) {
                const size_t flat = (size_t)(m - r) * Np + 2 * (size_t)n;
                const int u = r * K1;
                const double sgn = ((bin & 1) && (r & 1)) ? -1. : 1.;
                for (int k = 1; k <= L; ++k)
                    if (af[u + k])
                        acc_pair(&a, ar[u + k], ai[u + k], sr[flat - k], si[flat - k],
                                 sgn * sr[flat + k], sgn * si[flat + k]);
                if (af[u]) acc_mul(&a, ar[u], ai[u], sr[flat], si[flat]);
            }
            project_and_mirror(cr, ci, n, a, target, F, L);
        }
    }
}

void lwso_sweep(int flavour, double *sr, double *si, const double *wr, const double *wi,
                const int *wflag, const double *amp, int F, int M, int M0, int L, int Q, int Qp,
                double threshold, int update, double qdiv) {
    if (flavour == LWSO_FLAVOUR_NOFUTURE_Q4_COMPAT)
        sweep_nofuture_q4_compat(sr, si, wr, wi, wflag, amp, F, M, L, threshold);
    else
        sweep_canonical(sr, si, wr, wi, wflag, amp, F, M, M0, L, Q, Qp, threshold, update, qdiv);
}

/* ---- timing helpers (bench.py's cpu_baseline leg): `sweeps` dense sweeps in one call, so that a timed thread never
 * touches the Python interpreter between sweeps.  `fn` is a batch kernel with the calling convention of lwslib.h:9-12
 * (LWSQ2 / LWSQ4: has_q = 0; LWSanyQ / LWSfractionalQ: has_q = 1) -- the reference's own, out of oracle/_ref. */
typedef void (*lwso_batch_fn)(double *, double *, double *, double *, int *, double *, int, int, int, double);
typedef void (*lwso_batch_q_fn)(double *, double *, double *, double *, int *, double *, int, int, int, int, double);
void lwso_repeat_kernel(void *fn, int has_q, double *sr, double *si, double *wr, double *wi, int *wflag, double *amp,
                        int F, int M, int L, int Q, double threshold, int sweeps) {
    for (int i = 0; i < sweeps; ++i) {
        if (has_q) ((lwso_batch_q_fn)fn)(sr, si, wr, wi, wflag, amp, F, M, L, Q, threshold);
        else ((lwso_batch_fn)fn)(sr, si, wr, wi, wflag, amp, F, M, L, threshold);
    }
}
void lwso_repeat_sweep(double *sr, double *si, const double *wr, const double *wi, const int *wflag, const double *amp,
                       int F, int M, int L, int Q, int Qp, double threshold, int sweeps) {
    for (int i = 0; i < sweeps; ++i)
        sweep_canonical(sr, si, wr, wi, wflag, amp, F, M, LWSO_M0_ALL, L, Q, Qp, threshold, 2, (double)Q);
}

/* ---- online driver (lwslib.cpp:1424-1492) ---- */

void lwso_online(double *sr, double *si, const double *wr, const double *wi, const int *wflag,
                 const double *wr_ai, const double *wi_ai, const int *wflag_ai,
                 const double *wr_af, const double *wi_af, const int *wflag_af,
                 const double *amp, int iters, int LA, int F, int T, int L, int Q, int Qp,
                 double qdiv, const double *thr, int update) {
    const int Np = F + 2 * L;
    for (int m = 0; m < T; ++m) {
        int first = m - LA, count = LA; /* look-ahead window: frames first .. first+count-1, then m */
        if (first < 0) { first = 0; count = m; }
        const size_t om = (size_t)m * Np, of = (size_t)first * Np;
        /* newest frame: first estimate from the past only */
        sweep_canonical(sr + om, si + om, wr_ai, wi_ai, wflag_ai, amp + om, F, 1, 0, L, Q, Qp, 0.,
                        update, qdiv);
        for (int h = 0; h < iters; ++h) {
            if (LA > 0)
                sweep_canonical(sr + of, si + of, wr, wi, wflag, amp + of, F, count, count + 1, L,
                                Q, Qp, thr[h], update, qdiv);
            sweep_canonical(sr + om, si + om, wr_af, wi_af, wflag_af, amp + om, F, 1, 1, L, Q, Qp,
                            thr[h], update, qdiv);
        }
    }
}

/* ---- wrapper level (python/lws.pyx:209-375) ---- */

typedef struct {
    int T, F, L, Q, Qp, Np, Tp;
    double *er, *ei, *amp;
    double mean_amp;
} prep_t;

static void split_weights(const double *W, int count, double **wr, double **wi, int **wf) {
    *wr = (double *)malloc(sizeof(double) * count);
    *wi = (double *)malloc(sizeof(double) * count);
    *wf = (int *)malloc(sizeof(int) * count);
    for (int i = 0; i < count; ++i) {
        (*wr)[i] = W[2 * i];
        (*wi)[i] = W[2 * i + 1];
        /* lws.pyx:231-232: abs(W) > 1e-12, numpy abs == hypot */
        (*wf)[i] = hypot(W[2 * i], W[2 * i + 1]) > 1.0e-12;
    }
}

static void prep_make(prep_t *p, const double *S, int T, int F, int L, int Q, int Qp, double mean_amp) {
    p->T = T; p->F = F; p->L = L; p->Q = Q; p->Qp = Qp;
    p->Np = F + 2 * L; p->Tp = T + 2 * (Q - 1);
    const size_t n_in = (size_t)T * F, n_ext = (size_t)p->Tp * p->Np;
    double *ir = (double *)malloc(sizeof(double) * n_in), *ii = (double *)malloc(sizeof(double) * n_in);
    for (size_t i = 0; i < n_in; ++i) { ir[i] = S[2 * i]; ii[i] = S[2 * i + 1]; }
    p->er = (double *)malloc(sizeof(double) * n_ext);
    p->ei = (double *)malloc(sizeof(double) * n_ext);
    p->amp = (double *)malloc(sizeof(double) * n_ext);
    lwso_extend(p->er, p->ei, ir, ii, F, T, L, Q);
    for (size_t i = 0; i < n_ext; ++i) p->amp[i] = hypot(p->er[i], p->ei[i]); /* np.abs, lws.pyx:239 */
    if (mean_amp < 0) {
        double s = 0;
        for (size_t i = 0; i < n_in; ++i) s += hypot(ir[i], ii[i]);
        mean_amp = s / (double)n_in;
    }
    p->mean_amp = mean_amp;
    free(ir); free(ii);
}

static void prep_finish(prep_t *p, double *S_out) {
    const size_t n_in = (size_t)p->T * p->F;
    double *orr = (double *)malloc(sizeof(double) * n_in), *oi = (double *)malloc(sizeof(double) * n_in);
    lwso_extract(p->er, p->ei, orr, oi, p->F, p->T, p->L, p->Q);
    for (size_t i = 0; i < n_in; ++i) { S_out[2 * i] = orr[i]; S_out[2 * i + 1] = oi[i]; }
    free(orr); free(oi); free(p->er); free(p->ei); free(p->amp);
}

int lwso_batch_lws(const double *S_in, double *S_out, int T, int F, const double *W, int L, int Q,
                   int Qp, const double *thresholds, int iters, double mean_amp) {
    if (F % 2 == 0) return 1;
    if (iters == 0) { memcpy(S_out, S_in, sizeof(double) * 2 * (size_t)T * F); return 0; }
    prep_t p; double *wr, *wi; int *wf;
    prep_make(&p, S_in, T, F, L, Q, Qp, mean_amp);
    split_weights(W, Qp * Q * (L + 1), &wr, &wi, &wf);
    for (int i = 0; i < iters; ++i)
        sweep_canonical(p.er, p.ei, wr, wi, wf, p.amp, F, T, LWSO_M0_ALL, L, Q, Qp,
                        thresholds[i] * p.mean_amp, 2, (double)Q);
    prep_finish(&p, S_out);
    free(wr); free(wi); free(wf);
    return 0;
}

int lwso_nofuture_lws(const double *S_in, double *S_out, int T, int F, const double *W, int L,
                      int Q, int Qp, const double *thresholds, int iters, double mean_amp,
                      int nofuture_q4_compat) {
    if (F % 2 == 0) return 1;
    if (iters == 0) { memcpy(S_out, S_in, sizeof(double) * 2 * (size_t)T * F); return 0; }
    prep_t p; double *wr, *wi; int *wf;
    prep_make(&p, S_in, T, F, L, Q, Qp, mean_amp);
    split_weights(W, Qp * Q * (L + 1), &wr, &wi, &wf);
    const int compat = nofuture_q4_compat && Q == 4 && Qp == 4;
    for (int i = 0; i < iters; ++i) {
        const double thr = thresholds[i] * p.mean_amp;
        if (compat) sweep_nofuture_q4_compat(p.er, p.ei, wr, wi, wf, p.amp, F, T, L, thr);
        else sweep_canonical(p.er, p.ei, wr, wi, wf, p.amp, F, T, 0, L, Q, Qp, thr, 2, (double)Q);
    }
    prep_finish(&p, S_out);
    free(wr); free(wi); free(wf);
    return 0;
}

int lwso_online_lws(const double *S_in, double *S_out, int T, int F, const double *W,
                    const double *W_ai, const double *W_af, int L, int Q, int Qp,
                    const double *thresholds, int iters, int LA, double qdiv, double mean_amp) {
    if (F % 2 == 0) return 1;
    if (iters == 0) { memcpy(S_out, S_in, sizeof(double) * 2 * (size_t)T * F); return 0; }
    prep_t p; double *wr, *wi, *wr1, *wi1, *wr2, *wi2; int *wf, *wf1, *wf2;
    prep_make(&p, S_in, T, F, L, Q, Qp, mean_amp);
    const int nw = Qp * Q * (L + 1);
    split_weights(W, nw, &wr, &wi, &wf);
    split_weights(W_ai, nw, &wr1, &wi1, &wf1);
    split_weights(W_af, nw, &wr2, &wi2, &wf2);
    double *thr = (double *)malloc(sizeof(double) * iters);
    for (int i = 0; i < iters; ++i) thr[i] = thresholds[i] * p.mean_amp; /* lws.pyx:361 */
    lwso_online(p.er, p.ei, wr, wi, wf, wr1, wi1, wf1, wr2, wi2, wf2, p.amp, iters, LA, F, T, L, Q,
                Qp, qdiv, thr, 2);
    prep_finish(&p, S_out);
    free(thr);
    free(wr); free(wi); free(wf); free(wr1); free(wi1); free(wf1); free(wr2); free(wi2); free(wf2);
    return 0;
}
