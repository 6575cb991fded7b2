// lws_band_host.h -- host side of the band engine that needs no HIP: the geometry of a call and the two tables a kernel reads.
// Shared by lws_band.hip (the launcher) and tests/band_emul.cpp (the CPU emulation of the schedule).
#pragma once
#include "lws_band_core.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace lws {
namespace band {

// LT: the stencil half-width the kernel is compiled for (>= the plan's L).  SKW >= LT + 2: the value a lane reads from its
// left-hand neighbour is requested one step early and must have been written two steps before that.
inline Geom geometry(int F, int T, int Q, int LT, int SKW, int nls, int Pt, int helpers = 0) {
    Geom g{};
    g.F = F; g.T = T; g.Q = Q; g.SKW = SKW; g.nls = nls; g.Pt = Pt < 1 ? 1 : Pt;
    for (g.lg = 0; (1 << g.lg) < nls; ++g.lg) {}
    const int Tp = T + 2 * (Q - 1);
    // steps a lane spends on a frame: its F bins, LT steps before them (positions arrive LT bins ahead); the LT images above
    // Nyquist are written during the first steps of the lane's NEXT frame, which have no bin of their own
    g.P = (std::max(nls * SKW, F + LT) + SKW - 1) / SKW * SKW;
    g.gap = g.P - nls * SKW;
    // a sweep slot follows the one before it at the distance of the farthest old value it reads, + 2 (requested a step early)
    // (helper waves run a step ahead of their slot's main wave: what they read of the slot before is a step younger)
    g.LAG = std::max(LT + SKW * (Q - 1) + g.gap + 2 + (helpers > 0 ? 1 : 0), 3 * LT);   // (3 LT: an image above Nyquist is read 2 LT rows back)
    g.LAG += g.LAG & 1;                                         // (slots start on even steps: the step loop is unrolled by PFD = 2)
    g.R = g.LAG - LT + 1;
    g.nblk = (Tp + nls - 1) / nls;
    // steps of one sweep: until the last of the Tp frames is finished -- frame me sits in lane me % nls, block me / nls, and ends at step
    // SKW (me % nls) + P (me / nls + 1) -- + LT for its images.  (The last block is rarely full: its idle lanes are not waited for.)
    {
        const int rem = Tp - (g.nblk - 1) * nls;                // frames of the last block (1 .. nls)
        const long end_last = (long)SKW * (rem - 1) + (long)g.P * g.nblk;
        const long end_prev = g.nblk > 1 ? (long)SKW * (nls - 1) + (long)g.P * (g.nblk - 1) : 0;
        g.U = (int)std::max(end_last, end_prev) + LT + 3;
    }
    g.U += g.U & 1;
    // rows after the last one a step writes: the first slot's prefetch of step ux <= U - 1 + PFD reads row ux + LT + SKW r + gap
    g.rows = (long)g.U + PFD + LT + (long)SKW * (Q - 1) + g.gap + 8;
    return g;
}
inline size_t ring_bytes(const Geom &g, size_t csize) { return (size_t)g.R * g.nls * csize; }
// the mailboxes of a slot's helper waves: [2][helpers][nls][2] complex values
inline size_t mail_bytes(const Geom &g, int helpers, size_t csize) { return (size_t)2 * helpers * g.nls * 2 * csize; }
inline size_t table_bytes(const Geom &g, int LT, size_t csize) { return ((size_t)g.Q * (LT + 1) + (size_t)g.Pt * (g.Q - 1)) * csize; }   // weights + twiddles

// exp(2 pi j num / den), exact on the axes
inline void unit(long long num, long long den, double *re, double *im) {
    num %= den;
    if (num < 0) num += den;
    if ((4 * num) % den == 0) {
        static const double cs[4] = {1, 0, -1, 0}, sn[4] = {0, 1, 0, -1};
        const int q = (int)(4 * num / den);
        *re = cs[q]; *im = sn[q];
        return;
    }
    const double ang = 2.0 * M_PI * (double)num / (double)den;
    *re = std::cos(ang); *im = std::sin(ang);
}

// W: the plan's tensor on the host (complex128 interleaved, [Qp][Q][L+1]) with the twiddle structure (Pt, s) weights_twiddle()
// found.  wt: [Q][LT+1] complex (interleaved doubles): row 0 = W[0][0][k] (the frame's own taps; [0][0] is never read), row r =
// V[r][k] = W[0][r][k] exp(2 pi j r s k / Pt); weights the reference skips (|w| <= 1e-12, lws.pyx:231-232) and columns k > L are 0.
// tw: [Pt][Q-1]: exp(2 pi j p r s / Pt).
inline void tables(const double *W, int Q, int L, int LT, int Pt, int s, std::vector<double> &wt, std::vector<double> &tw) {
    const int K1 = L + 1, KT = LT + 1;
    if (Pt < 1) { Pt = 1; s = 0; }
    wt.assign((size_t)Q * KT * 2, 0.0);
    for (int r = 0; r < Q; ++r)
        for (int k = 0; k <= L; ++k) {
            if (r == 0 && k == 0) continue;
            const double br = W[2 * ((size_t)r * K1 + k)], bi = W[2 * ((size_t)r * K1 + k) + 1];
            if (!(std::hypot(br, bi) > 1.0e-12)) continue;
            double cr, ci;
            unit((long long)r * s * k, Pt, &cr, &ci);
            wt[2 * ((size_t)r * KT + k)] = br * cr - bi * ci;
            wt[2 * ((size_t)r * KT + k) + 1] = br * ci + bi * cr;
        }
    tw.assign((size_t)Pt * (Q - 1) * 2, 0.0);
    for (int p = 0; p < Pt; ++p)
        for (int r = 1; r < Q; ++r)
            unit((long long)p * r * s, Pt, &tw[2 * ((size_t)p * (Q - 1) + r - 1)], &tw[2 * ((size_t)p * (Q - 1) + r - 1) + 1]);
}

}  // namespace band
}  // namespace lws
