// lws_team.hip -- the team engine: online (TF_RTISI_LA, lwslib.cpp:1424-1492) and no-future (lwslib.cpp:620-764) sweeps of the
// shapes the LDS engines do not take -- more than 8 frames per stencil row, stencils wider than L = 5, weights without the
// twiddle structure, frames whose ring does not fit the LDS -- in fp32 and fp64.
//
// The generic engine (lws_generic.hip) replays the reference's sequential sweep as a skewed wavefront, one LANE per bin: with
// Q = 16 a lane walks through (2Q-1)(2L+1) = 341 taps on its own while the 20-odd bins of a step leave 1000 lanes idle, and a
// step -- there are half a million in config 3's online stage at hop = frame/16 -- takes 40 us.  Here a bin belongs to a TEAM
// of G lanes (a power of two, as many as the step's units leave room for in a workgroup of 1024): lane g takes terms g, g + G,
// ... of the bin's sum, all its loads are in flight together, the partial sums are combined by a butterfly over the team and
// lane 0 re-projects and writes.  Same schedule  t = D s + (L+1) m + c,  D = Q (L+1)  (lws_generic.hip, head), same sweeps in
// the reference's call order, same arithmetic per tap (the grouped form of lwslib.cpp:310-311); only the order in which a
// bin's terms are added differs from the sequential loop: rounding-level, deterministic.
//
// A TERM is one statement of the reference's tap loop -- a += w b + conj(w) c with b, c the two mirrored neighbours, or one
// of them when the frame on the right does not take part (rframe, lwslib.cpp:1143-1151) -- described once per launch in an
// LDS table: element offsets of b and c from the bin, index of the weight in its row, which of the two rows (c mod Qp or its
// negative) it comes from, frame distance.  `mac` and `macc` of the generic engine are the same statement with c = 0 / b = 0
// (x + 0 and x - 0 are exact), so one body serves every term.
//
// Three kernels for the online driver: k_team_online_ring (fp32 plans: the moving window, targets and weights in LDS; placement of a
// lane's terms per sweep), k_team_online (the same sums with the state in memory, when the ring does not fit), and
// k_team_online_ordered (fp64 plans: the increments by the team, the sum by ONE lane in the reference's order -- the generic
// engine's bits).  k_team_sweeps: no-future sweeps.  Pins: LWS_TEAM_LANES=1 makes every kernel add in the generic engine's order (its
// bits); tools/stress_team.py compares random shapes bit for bit; LWS_TEAM_DBG_POISON=1 starts the ring kernel's LDS as NaNs.
#include "lws_team.h"

#include <cstdlib>
#include <type_traits>

namespace lws {
namespace {

template <typename real> struct Chunk { static constexpr int N = sizeof(real) == 8 ? 4 : 8; };   // terms of a lane whose loads are in flight together

template <typename real, typename C>
__device__ __forceinline__ void pair(C &a, const C w, const C b, const C c) {   // lws_generic.hip: pair
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}

// term table entry: {offset of b, offset of c (elements, from the bin), weight index in its row,
//                    r | negrow << 8 | (what !both removes: 1 = c, 2 = b) << 9 | centre << 11}
struct Term { int ob, oc, wi, meta; };

// entries [0, L): the centre frame's k = 1..L; then per r = 1..Q-1: (r, 0) and, for k = 1..L, (r, k) of the row, (r, k) of the negative row:
// the order of the statements in lws_generic.hip's accumulate_v, so that ONE lane per bin (LWS_TEAM_LANES=1) adds them in that engine's order
__device__ __forceinline__ void build_terms(Term *tt, int NT, int L, int Np, int tid, int nthr) {
    const int K1 = L + 1, W21 = 2 * L + 1;
    for (int j = tid; j < NT; j += nthr) {
        Term e;
        if (j < L) {
            const int k = j + 1;
            e.ob = -k; e.oc = k; e.wi = k; e.meta = 1 << 11;
        } else {
            const int jj = j - L, r = 1 + jj / W21, qk = jj - (r - 1) * W21, u = r * K1, k = (qk + 1) / 2;
            if (qk == 0) { e.ob = -r * Np; e.oc = r * Np; e.wi = u; e.meta = r | (1 << 9); }
            else if (qk & 1) { e.ob = -r * Np - k; e.oc = r * Np - k; e.wi = u + k; e.meta = r | (1 << 9); }
            else { e.ob = r * Np + k; e.oc = -r * Np + k; e.wi = u + k; e.meta = r | (1 << 8) | (2 << 9); }
        }
        tt[j] = e;
    }
}

template <typename real> __device__ __forceinline__ real shfl_xor(real v, int off) { return __shfl_xor(v, off, 64); }

// the terms of a lane's first chunk, read from the table once per launch (they do not change from step to step)
template <typename real> struct LaneTerms { Term e[Chunk<real>::N]; };
template <typename real>
__device__ __forceinline__ LaneTerms<real> lane_terms(const Term *tt, int NT, int g, int G) {
    LaneTerms<real> lt;
#pragma unroll
    for (int i = 0; i < Chunk<real>::N; ++i) {
        const int j = g + i * G;
        lt.e[i] = tt[j < NT ? j : 0];
        if (j >= NT) lt.e[i].meta = 1 << 12;   // no such term
    }
    return lt;
}

// One bin by one team.  `unit`: the team has a bin in this step (uniform over the team).  Every lane of the workgroup calls this
// in every step (the butterfly needs whole waves).  The target magnitude, the weights and the taps are requested together --
// one memory round trip per step -- so the sum is formed whether or not the bin turns out to be active (lwslib.cpp:84-85);
// only an active bin is written.
template <typename real>
__device__ __forceinline__ void team_bin(typename cx<real>::type *S, const real *amp, const Term *tt, const LaneTerms<real> &lt, int NT, bool unit,
                                         int m_ext, int c, bool centre, int two_sided, const WeightSet<real> ws, real thr, int F, int L, int Q,
                                         int Qp, bool add_self, real qdiv, int g, int G) {
    using C = typename cx<real>::type;
    constexpr int CH = Chunk<real>::N;
    const int Np = F + 2 * L, RQ = Q * (L + 1);
    const int n = c + L;
    const size_t idx = unit ? (size_t)m_ext * Np + n : 0;
    C a;
    a.x = 0; a.y = 0;
    C *ctr = S + idx;
    real target = 0;
    if (unit) {
        target = amp[idx];
        const int row = c % Qp, rowneg = (Qp - row) % Qp;
        const C *w0 = ws.w + (size_t)row * RQ, *w1 = ws.w + (size_t)rowneg * RQ;
        // (no flag reads: the device copy of a tensor holds exact zeros where the reference skips a weight, |w| <= 1e-12 -- lws_capi.hip:
        //  upload_weights -- and a zero weight adds 0 (b +- c) = +-0 to a sum that is never -0: the same value as skipping)
        if (g == 0 && centre && add_self) { const C s0 = ctr[0]; a.x += s0.x / qdiv; a.y += s0.y / qdiv; }
        for (int j0 = g; j0 < NT; j0 += G * CH) {
            C w[CH], vb[CH], vc[CH];
            bool live[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                Term e = lt.e[i];
                if (j0 != g) {   // (further chunks: thin teams only)
                    const int j = j0 + i * G;
                    e = tt[j < NT ? j : 0];
                    if (j >= NT) e.meta = 1 << 12;
                }
                const int r = e.meta & 0xff, neg = (e.meta >> 8) & 1, cut = (e.meta >> 9) & 3;
                const bool is_centre = (e.meta >> 11) & 1, none = (e.meta >> 12) & 1;
                const bool both = is_centre || r < two_sided;
                live[i] = !none && (is_centre ? centre : true);
                w[i] = (neg ? w1 : w0)[e.wi];
                const C xb = ctr[e.ob], xc = ctr[e.oc];   // (always inside the extended buffer: Q - 1 pad frames, L pad columns)
                C z; z.x = 0; z.y = 0;
                vb[i] = (!both && cut == 2) ? z : xb;
                vc[i] = (!both && cut == 1) ? z : xc;
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                C t = a;
                pair<real>(t, w[i], vb[i], vc[i]);
                a = live[i] ? t : a;
            }
        }
    }
    const bool act = unit && (target > thr);
    // butterfly over the team (teams are aligned groups of G lanes of a wave): every lane ends with the same sum
    if (__any(act)) {
        for (int off = 1; off < G; off <<= 1) {
            const real ox = shfl_xor<real>(a.x, off), oy = shfl_xor<real>(a.y, off);
            a.x += ox; a.y += oy;
        }
    }
    if (act && g == 0) {
        const real mag = sqrt(a.x * a.x + a.y * a.y);
        if (mag > 0) {
            C v;
            v.x = a.x * target / mag;
            v.y = a.y * target / mag;
            ctr[0] = v;
            // Hermitian images in the pad columns (lwslib.cpp:362-367)
            const int nyq = F + L - 1;
            C vc;
            vc.x = v.x; vc.y = -v.y;
            if (n >= L + 1 && n < 2 * L + 1) S[(size_t)m_ext * Np + 2 * L - n] = vc;
            else if (n >= F - 1 && n < nyq) S[(size_t)m_ext * Np + 2 * nyq - n] = vc;
        }
    }
}

struct TeamGeom { int G, nsl, nunits; };

// online: sweep s = (frame m = s / per, q = s % per) is owned by slot s mod nsl; a slot has LA + 1 units (frame positions)
template <typename real>
__global__ void __launch_bounds__(1024) k_team_online(GenericArgs<real> a, TeamGeom tg) {
    using C = typename cx<real>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    Term *tt = reinterpret_cast<Term *>(tsm);
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    C *S = a.state + (size_t)b * Tp * Np;
    const real *amp = a.amp + (size_t)b * Tp * Np;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const int NT = L + (Q - 1) * (2 * L + 1);
    const bool add_self = (a.update == 1);
    build_terms(tt, NT, L, Np, tid, nthr);
    __syncthreads();
    const int G = tg.G, g = tid & (G - 1), team = tid / G;
    const LaneTerms<real> lt = lane_terms<real>(tt, NT, g, G);
    const int LA = a.LA, per = a.n_thr + 1, rps = LA + 1;
    const int slot = team / rps, j = team - slot * rps;
    const long nsweeps = (long)T * per;
    const long t_end = D * (nsweeps - 1) + (long)sk * (T - 1) + F;  // one past the last step (lws_generic.hip)
    long s = slot < tg.nsl ? slot : nsweeps;
    // the unit of this team in sweep s
    long t0 = 0, s_end = -1;
    int m_ext = 0, two_sided = 1, wsel = 0;
    bool valid = false, centre = false;
    real th = 0;
    auto setup = [&]() {
        valid = false;
        if (s >= nsweeps) { s_end = t_end; return; }
        const int m = (int)(s / per), q = (int)(s - (long)m * per);
        s_end = D * s + (long)sk * m + F - 1;                // last step of the sweep (its last frame is m)
        int first = m - LA;
        if (first < 0) first = 0;
        int rho;
        if (q == 0) { if (j != 0) return; rho = m; }        // first estimate of frame m from the past: W_ai, threshold 0
        else { rho = first + j; if (rho > m) return; }
        valid = true;
        t0 = D * s + (long)sk * rho;
        m_ext = rho + Q - 1;
        if (q == 0) { centre = false; two_sided = 1; wsel = 1; th = 0; }
        else {
            int ts = m - rho + 1;
            if (ts > Q) ts = Q;
            centre = true; two_sided = ts; wsel = (rho == m) ? 2 : 0; th = thr[q - 1];
        }
    };
    setup();
    for (long t = 0; t < t_end; ++t) {
        while (t > s_end) { s += tg.nsl; setup(); }
        const long cl = t - t0;
        const bool unit = valid && cl >= 0 && cl < F;
        team_bin<real>(S, amp, tt, lt, NT, unit, m_ext, (int)cl, centre, two_sided, a.w[wsel], th, F, L, Q, Qp, add_self, a.qdiv, g, G);
        __syncthreads();
    }
}

// ---- the online driver, order-exact: G lanes fetch and multiply, ONE lane adds ---------------------------------------------------
// What makes the generic engine slow is not its additions but the latency of 341 dependent loads per bin; what makes a re-associated
// sum unfit for an fp64 plan is the recursion, which amplifies its rounding by 5-10 per frame until the phases -- equally consistent --
// are no longer the reference's.  So the fp64 plans' variant keeps the order: a team's lanes form the INCREMENTS of a bin's terms
// (the value each statement of the reference's tap loop adds: w b + conj(w) c, every product and difference rounded as there) and
// leave them in LDS, [unit][term]; after a barrier one lane per bin adds them in the reference's order -- 2 NT dependent additions,
// all bins of the step side by side in one wave -- re-projects and writes.  Same bits as lws_generic.hip (a skipped term is a +0.0:
// a sum that starts at +0.0 never is -0.0), two barriers per step.
template <typename real> struct OrdUnit { long long idx; real target; int act, n, m_ext, pad; };

template <typename real>
__global__ void __launch_bounds__(1024) k_team_online_ordered(GenericArgs<real> a, TeamGeom tg, unsigned off_inc, unsigned off_unit, int NTP) {
    using C = typename cx<real>::type;
    constexpr int CH = Chunk<real>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    Term *tt = reinterpret_cast<Term *>(tsm);
    C *inc = reinterpret_cast<C *>(tsm + off_inc);                        // [unit][NTP]: slot 0 the S / qdiv term, slot 1 + j term j
    OrdUnit<real> *ui = reinterpret_cast<OrdUnit<real> *>(tsm + off_unit);
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), RQ = Q * (L + 1);
    C *S = a.state + (size_t)b * Tp * Np;
    const real *amp = a.amp + (size_t)b * Tp * Np;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const int NT = L + (Q - 1) * (2 * L + 1);
    const bool add_self = (a.update == 1);
    build_terms(tt, NT, L, Np, tid, nthr);
    for (int i = tid; i < tg.nunits; i += nthr) ui[i].act = 0;
    __syncthreads();
    const int G = tg.G, g = tid & (G - 1), team = tid / G;
    const int LA = a.LA, per = a.n_thr + 1, rps = LA + 1;
    const int slot = team / rps, j = team - slot * rps;
    const long nsweeps = (long)T * per;
    const long t_end = D * (nsweeps - 1) + (long)sk * (T - 1) + F;
    long s = slot < tg.nsl ? slot : nsweeps;
    long t0 = 0, s_end = -1;
    int m_ext = 0, two_sided = 1, wsel = 0;
    bool valid = false, centre = false;
    real th = 0;
    auto setup = [&]() __attribute__((always_inline)) {
        valid = false;
        if (s >= nsweeps) { s_end = t_end; return; }
        const int m = (int)(s / per), q = (int)(s - (long)m * per);
        s_end = D * s + (long)sk * m + F - 1;
        int first = m - LA;
        if (first < 0) first = 0;
        int rho;
        if (q == 0) { if (j != 0) return; rho = m; }
        else { rho = first + j; if (rho > m) return; }
        valid = true;
        t0 = D * s + (long)sk * rho;
        m_ext = rho + Q - 1;
        if (q == 0) { centre = false; two_sided = 1; wsel = 1; th = 0; }
        else {
            int ts = m - rho + 1;
            if (ts > Q) ts = Q;
            centre = true; two_sided = ts; wsel = (rho == m) ? 2 : 0; th = thr[q - 1];
        }
    };
    setup();
    for (long t = 0; t < t_end; ++t) {
        while (t > s_end) { s += tg.nsl; setup(); }
        const long cl = t - t0;
        const bool unit = valid && cl >= 0 && cl < F && team < tg.nunits;
        // ---- the increments
        if (unit) {
            const int c = (int)cl, n = c + L;
            const size_t idx = (size_t)m_ext * Np + n;
            const C *ctr = S + idx;
            const real target = amp[idx];
            const int row = c % Qp, rowneg = (Qp - row) % Qp;
            const WeightSet<real> ws = a.w[wsel];
            const C *w0 = ws.w + (size_t)row * RQ, *w1 = ws.w + (size_t)rowneg * RQ;
            C *mine = inc + (size_t)team * NTP;
            if (g == 0) {
                C pre; pre.x = 0; pre.y = 0;
                if (centre && add_self) { const C s0 = ctr[0]; pre.x = s0.x / a.qdiv; pre.y = s0.y / a.qdiv; }
                mine[0] = pre;
                OrdUnit<real> u;
                u.idx = (long long)idx; u.target = target; u.act = (target > th) ? 1 : 0; u.n = n; u.m_ext = m_ext; u.pad = 0;
                ui[team] = u;
            }
            for (int j0 = g; j0 < NT; j0 += G * CH) {
                C w[CH], vb[CH], vc[CH];
                bool live[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int jj = j0 + i * G;
                    const Term e = tt[jj < NT ? jj : 0];
                    const int r = e.meta & 0xff, neg = (e.meta >> 8) & 1, cut = (e.meta >> 9) & 3;
                    const bool is_centre = (e.meta >> 11) & 1;
                    const bool both = is_centre || r < two_sided;
                    live[i] = jj < NT && (is_centre ? centre : true);   // (a weight the reference skips is an exact zero in the device copy)
                    w[i] = (neg ? w1 : w0)[e.wi];
                    const C xb = ctr[e.ob], xc = ctr[e.oc];
                    C z; z.x = 0; z.y = 0;
                    vb[i] = (!both && cut == 2) ? z : xb;
                    vc[i] = (!both && cut == 1) ? z : xc;
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int jj = j0 + i * G;
                    C v; v.x = 0; v.y = 0;
                    pair<real>(v, w[i], vb[i], vc[i]);            // 0 + increment: the increment
                    if (!live[i]) { v.x = 0; v.y = 0; }
                    if (jj < NT) mine[1 + jj] = v;
                }
            }
        } else if (g == 0 && team < tg.nunits) ui[team].act = 0;
        __syncthreads();
        // ---- one lane per bin: the sum in the reference's order, the re-projection, the write
        if (tid < tg.nunits) {
            const OrdUnit<real> u = ui[tid];
            if (u.act) {
                const C *mine = inc + (size_t)tid * NTP;
                C acc;
                acc.x = 0; acc.y = 0;
                for (int jj = 0; jj <= NT; ++jj) { const C v = mine[jj]; acc.x += v.x; acc.y += v.y; }
                const real mag = sqrt(acc.x * acc.x + acc.y * acc.y);
                if (mag > 0) {
                    C v;
                    v.x = acc.x * u.target / mag;
                    v.y = acc.y * u.target / mag;
                    S[u.idx] = v;
                    const int n = u.n, nyq = F + L - 1;   // Hermitian images in the pad columns (lwslib.cpp:362-367)
                    C vc;
                    vc.x = v.x; vc.y = -v.y;
                    if (n >= L + 1 && n < 2 * L + 1) S[(size_t)u.m_ext * Np + 2 * L - n] = vc;
                    else if (n >= F - 1 && n < nyq) S[(size_t)u.m_ext * Np + 2 * nyq - n] = vc;
                }
            }
        }
        __syncthreads();
    }
}

// ---- the online driver with its moving window in LDS ------------------------------------------------------------------------
// What bounds k_team_online is the texture path: 2 NT + NT scattered loads (taps, weights) per bin, 4 us a step.
// TF_RTISI_LA touches a short window -- the sweeps in flight read extended frames m_lo - LA .. m_hi + Q - 1 -- so a ring of
// NWR = Q + LA + DM extended frames (DM: how far apart the online frames of the sweeps in flight can be) is kept in LDS together
// with the target magnitudes of the frames that can still change, and the three weight tensors (summarised ones;
// general tensors stay in memory): HBM sees a frame once on its way in and once on its way out, a step's loads are LDS reads.
struct RingGeom { int NWR, NWA, DM, wl; unsigned off_ring, off_amp, off_w, bytes; int poison, fits; };

// a term of the ring kernel in one word: r | negrow << 8 | (what !both removes) << 9 | centre << 11 | none << 12 | (dk + 32) << 13,
// dk the column offset of b (c: the same column, the mirrored one for the centre frame); the weight index is r (L+1) + |dk|
__device__ __forceinline__ void build_terms_ring(int *tt, int NT, int L, int tid, int nthr) {
    const int W21 = 2 * L + 1;
    for (int j = tid; j < NT; j += nthr) {
        int meta, dk;
        if (j < L) { dk = -(j + 1); meta = 1 << 11; }
        else {
            const int jj = j - L, r = 1 + jj / W21, qk = jj - (r - 1) * W21;
            if (qk == 0) { dk = 0; meta = r | (1 << 9); }
            else if (qk & 1) { dk = -((qk + 1) / 2); meta = r | (1 << 9); }
            else { dk = qk / 2; meta = r | (1 << 8) | (2 << 9); }
        }
        tt[j] = meta | ((dk + 32) << 13);
    }
}

// NCH: chunks of four terms whose placement a lane keeps in registers -- two, or three in a workgroup of at most 512 threads (twice the
// registers per lane): sixteen lanes a bin instead of thirty-two at hop = frame/16
template <typename real, bool WL, int NCH>
__global__ void __launch_bounds__(NCH == 3 ? 512 : 1024) k_team_online_ring(GenericArgs<real> a, TeamGeom tg, RingGeom rg) {
    using C = typename cx<real>::type;
    constexpr int CH = 4;   // terms of a lane in flight together (LDS round trips are short: small chunks, few registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    int *tt = reinterpret_cast<int *>(tsm);
    C *ring = reinterpret_cast<C *>(tsm + rg.off_ring);        // NWR frames and one row of zeros (what a term that does not take part reads)
    real *ampr = reinterpret_cast<real *>(tsm + rg.off_amp);
    C *wl = reinterpret_cast<C *>(tsm + rg.off_w);             // WL: the three tensors (exact zeros where the reference skips a weight)
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), RQ = Q * (L + 1);
    C *gS = a.state + (size_t)b * Tp * Np;
    const real *gA = a.amp + (size_t)b * Tp * Np;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const int NT = L + (Q - 1) * (2 * L + 1);
    const bool add_self = (a.update == 1);
    const int NWR = rg.NWR, NWA = rg.NWA;
    const int zrow = NWR * Np + L;                              // (column 0 of the zero row: every column offset -L .. L stays inside)
    if (rg.poison) {   // (LWS_TEAM_DBG_POISON=1, tests: the whole allocation and a 16 KB tail the launcher adds behind it start as NaNs --
                       //  what the kernel reads without having written it, inside its regions or past their end, shows in the result)
        for (unsigned i = tid; i < rg.bytes / 4 + 4096; i += nthr) reinterpret_cast<float *>(tsm)[i] = __builtin_nanf("");
        __syncthreads();
    }
    build_terms_ring(tt, NT, L, tid, nthr);
    for (int i = tid; i < Np; i += nthr) { C z; z.x = 0; z.y = 0; ring[NWR * Np + i] = z; }
    if constexpr (WL) {
        // A weight without a flag (|w| <= 1e-12, lws.pyx:227-232) is skipped by the reference; the device copy holds an exact zero there
        // (upload_weights), which adds w b = 0 to a finite sum: the same value.
        for (int set = 0; set < 3; ++set)
            for (int i = tid; i < Qp * RQ; i += nthr) {
                wl[set * Qp * RQ + i] = a.w[set].w[i];
            }
    }
    // ---- the ring: extended frame e lives in row e mod NWR (targets: e mod NWA); frames [0, loaded) have been brought in
    int loaded = 0;
    auto bring = [&]() __attribute__((always_inline)) {   // the next frame in, the one it replaces out
        const int e = loaded, row = (e % NWR) * Np, rowa = (e % NWA) * Np;
        for (int i = tid; i < Np; i += nthr) {
            if (e >= NWR) gS[(size_t)(e - NWR) * Np + i] = ring[row + i];
            ring[row + i] = gS[(size_t)e * Np + i];
            ampr[rowa + i] = gA[(size_t)e * Np + i];
        }
        ++loaded;
    };
    const int e_last = T + Q - 2;                              // frame T-1; the pad frames on the right are never touched (rframe)
    while (loaded <= Q - 1 && loaded <= e_last) bring();       // the pad frames on the left and frame 0
    __syncthreads();
    const int G = tg.G, g = tid & (G - 1), team = tid / G;
    constexpr int NC = sizeof(real) == 8 ? CH : NCH * CH;      // terms of a lane whose placement is kept in registers (fp32: two or three chunks)
    int lt[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) { const int jj = g + i * G; lt[i] = jj < NT ? tt[jj] : (1 << 12); }
    const bool two_chunks = g + CH * G < NT;
    const int LA = a.LA, per = a.n_thr + 1, rps = LA + 1;
    const int slot = team / rps, j = team - slot * rps;
    const long nsweeps = (long)T * per;
    const long t_end = D * (nsweeps - 1) + (long)sk * (T - 1) + F;  // one past the last step (lws_generic.hip)
    const long t_frame = (long)D * per + sk;                   // the first sweep of online frame m starts at step t_frame m
    long t_bring = t_frame;                                    // ... and needs extended frame m + Q - 1 from then on (m = 1 next)
    long s = slot < tg.nsl ? slot : nsweeps;
    long t0 = 0, s_end = -1;
    int c = 0, c_end = -1, row = 0;                            // the unit's bin in this step (t - t0), its last step, c mod Qp
    int em = 0, ea = 0, two_sided = 1, wsel = 0;
    // what a sweep fixes for a term: where b and c are in the ring (the zero row if the term, or that side of it, does not take
    // part: rframe / cframe, lwslib.cpp:1143-1151), the weight's index in its row, which row (c mod Qp or its negative)
    int rb[NC], rc[NC], wn[NC];                                // (wn: index | negative row << 31)
    bool valid = false, centre = false;
    real th = 0;
    auto row_of = [&](int dr) __attribute__((always_inline)) { int x = em + dr; x += x < 0 ? NWR : 0; x -= x >= NWR ? NWR : 0; return x * Np; };
    auto place = [&](int meta, int &ob, int &oc, int &w_n) __attribute__((always_inline)) {
        const int r = meta & 0xff, cut = (meta >> 9) & 3, dk = ((meta >> 13) & 63) - 32;
        const bool is_centre = (meta >> 11) & 1, none = (meta >> 12) & 1;
        const bool both = is_centre || r < two_sided, dead = none || (is_centre && !centre);
        const int drb = cut == 1 ? -r : (cut == 2 ? r : 0);
        ob = (dead || (!both && cut == 2)) ? zrow : row_of(drb) + dk;
        oc = (dead || (!both && cut == 1)) ? zrow : row_of(-drb) + (cut == 0 ? -dk : dk);
        // (a term past the end has no weight of its own: entry 0 of the row, times the row of zeros.  Its index must stay inside the
        // tensor: beyond the LDS copy lies whatever the kernel before left there, and a NaN there times zero is a NaN -- a bin that is
        // then not written.  tools/stress_team.py found it: one bin in a few thousand, after fp64 cases had run on the same CU.)
        w_n = none ? 0 : ((r * (L + 1) + (dk < 0 ? -dk : dk)) | (((meta >> 8) & 1) << 31));
    };
    auto setup = [&]() __attribute__((always_inline)) {
        valid = false;
        if (s >= nsweeps) { s_end = t_end; return; }
        const int m = (int)(s / per), q = (int)(s - (long)m * per);
        s_end = D * s + (long)sk * m + F - 1;
        int first = m - LA;
        if (first < 0) first = 0;
        int rho;
        if (q == 0) { if (j != 0) return; rho = m; }
        else { rho = first + j; if (rho > m) return; }
        valid = true;
        t0 = D * s + (long)sk * rho;
        const int m_ext = rho + Q - 1;
        em = m_ext % NWR; ea = m_ext % NWA;
        if (q == 0) { centre = false; two_sided = 1; wsel = 1; th = 0; }
        else {
            int ts = m - rho + 1;
            if (ts > Q) ts = Q;
            centre = true; two_sided = ts; wsel = (rho == m) ? 2 : 0; th = thr[q - 1];
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) place(lt[i], rb[i], rc[i], wn[i]);
    };
    auto advance = [&](long t) __attribute__((always_inline)) {   // the sweep this team works on at step t
        while (t > s_end) { s += tg.nsl; setup(); }
        const long d = t - t0, de = s_end - t0;
        c = d < -(1 << 30) ? -(1 << 30) : (int)d;             // (a sweep that starts far ahead: its own advance() comes first)
        c_end = valid ? (int)(de < (1 << 30) ? de : (1 << 30)) : (1 << 30);
        if (!valid) c = -(1 << 30);
        row = c > 0 ? c % Qp : 0;
    };
    setup();
    advance(0);
    for (long t = 0; t < t_end; ++t) {
        if (t >= t_bring && loaded <= e_last) {                // (uniform over the workgroup)
            bring();
            t_bring += t_frame;
            __syncthreads();
        }
        if (t > s_end) advance(t);                             // (s_end of a lane without a sweep left: t_end)
        const bool unit = c >= 0 && c < F;
        const int n = c + L;
        C acc;
        acc.x = 0; acc.y = 0;
        real target = 0;
        if (unit) {
            target = ampr[ea * Np + n];
            const int rowneg = row == 0 ? 0 : Qp - row;
            const int d0 = (WL ? wsel * Qp + row : row) * RQ, dn = (rowneg - row) * RQ;   // weight rows: d0, d0 + dn
            const C *wg = WL ? nullptr : a.w[wsel].w;
            if (g == 0 && centre && add_self) { const C s0 = ring[em * Np + n]; acc.x += s0.x / a.qdiv; acc.y += s0.y / a.qdiv; }
            auto chunk = [&](auto off_c, const auto &ob, const auto &oc, const auto &w_n) {   // (arrays by reference, constant indices: registers)
                constexpr int O = decltype(off_c)::value;
                C w[CH], vb[CH], vc[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int wo = d0 + (w_n[O + i] & 0x7fffffff) + ((w_n[O + i] >> 31) & dn);
                    if constexpr (WL) w[i] = wl[wo];
                    else {
                        w[i] = wg[wo];
                    }
                    vb[i] = ring[ob[O + i] + n];
                    vc[i] = ring[oc[O + i] + n];
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) pair<real>(acc, w[i], vb[i], vc[i]);
            };
            chunk(std::integral_constant<int, 0>{}, rb, rc, wn);
            if constexpr (NC > CH) { if (two_chunks) chunk(std::integral_constant<int, CH>{}, rb, rc, wn); }
            if constexpr (NC > 2 * CH) { if (g + 2 * CH * G < NT) chunk(std::integral_constant<int, 2 * CH>{}, rb, rc, wn); }
            for (int j0 = g + G * NC; j0 < NT; j0 += G * CH) {   // further chunks (thin teams): their terms from the table
                int ob[CH], oc[CH], w_n[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int jj = j0 + i * G;
                    place(jj < NT ? tt[jj] : (1 << 12), ob[i], oc[i], w_n[i]);
                }
                chunk(std::integral_constant<int, 0>{}, ob, oc, w_n);
            }
        }
        const bool act = unit && (target > th);
        if (__any(act)) {
            for (int off = 1; off < G; off <<= 1) {
                const real ox = shfl_xor<real>(acc.x, off), oy = shfl_xor<real>(acc.y, off);
                acc.x += ox; acc.y += oy;
            }
        }
        if (act && g == 0) {
            const real mag = sqrt(acc.x * acc.x + acc.y * acc.y);
            if (mag > 0) {
                C v;
                v.x = acc.x * target / mag;
                v.y = acc.y * target / mag;
                C *fr = ring + em * Np;
                fr[n] = v;
                const int nyq = F + L - 1;   // Hermitian images in the pad columns (lwslib.cpp:362-367)
                C vc;
                vc.x = v.x; vc.y = -v.y;
                if (n >= L + 1 && n < 2 * L + 1) fr[2 * L - n] = vc;
                else if (n >= F - 1 && n < nyq) fr[2 * nyq - n] = vc;
            }
        }
        __syncthreads();
        ++c;                                                   // the next step's bin
        row = c > 0 ? (row + 1 == Qp ? 0 : row + 1) : 0;
    }
    // what is still in the ring goes back
    for (int e = loaded > NWR ? loaded - NWR : 0; e < loaded; ++e) {
        const int row = (e % NWR) * Np;
        for (int i = tid; i < Np; i += nthr) gS[(size_t)e * Np + i] = ring[row + i];
    }
}

// no-future (and batch in the reference's layout): `ng` sweeps in flight, frame j of a hyperplane per unit (lws_generic.hip:
// the MODE_BATCH / MODE_NOFUTURE / MODE_ASYM loop, same unit numbering)
template <typename real>
__global__ void __launch_bounds__(1024) k_team_sweeps(GenericArgs<real> a, TeamGeom tg) {
    using C = typename cx<real>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    Term *tt = reinterpret_cast<Term *>(tsm);
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    C *S = a.state + (size_t)b * Tp * Np;
    const real *amp = a.amp + (size_t)b * Tp * Np;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const int NT = L + (Q - 1) * (2 * L + 1);
    const bool add_self = (a.update == 1);
    build_terms(tt, NT, L, Np, tid, nthr);
    __syncthreads();
    const int G = tg.G, g = tid & (G - 1), team = tid / G;
    const LaneTerms<real> lt = lane_terms<real>(tt, NT, g, G);
    const WeightSet<real> ws = a.w[a.wsel];
    const bool asym = (a.mode == MODE_ASYM);
    const bool centre0 = (a.mode == MODE_BATCH);
    const int ts0 = (a.mode == MODE_BATCH) ? Q : 1;
    int lpi = (F - 1) / sk + 1;  // frames that can sit on one hyperplane
    if (lpi > T) lpi = T;
    const int group = tg.nsl;
    const int k = team / lpi, j = team - k * lpi;
    for (int g0 = 0; g0 < a.n_thr; g0 += group) {
        const int ng = (a.n_thr - g0 < group) ? a.n_thr - g0 : group;
        const int nsteps = F + sk * (T - 1) + D * (ng - 1);
        const real th = (k < ng) ? thr[g0 + k] : (real)0;
        for (int step = 0; step < nsteps; ++step) {
            bool unit = team < tg.nunits && k < ng;
            int mm = 0, c = 0, two_sided = ts0;
            bool centre = centre0;
            const int u = step - D * k;
            if (unit && u >= 0) {
                int jhi = u / sk;
                if (jhi > T - 1) jhi = T - 1;
                mm = jhi - j;
                c = u - sk * mm;
                unit = mm >= 0 && c < F;
                if (asym) {  // cframe / rframe of lwslib.cpp:1143-1151 for local frame mm
                    two_sided = a.M0 - mm;
                    if (two_sided > Q) two_sided = Q;
                    centre = two_sided >= 1;
                    if (two_sided < 1) two_sided = 1;
                }
            } else unit = false;
            team_bin<real>(S, amp, tt, lt, NT, unit, mm + Q - 1, c, centre, two_sided, ws, th, F, L, Q, Qp, add_self, a.qdiv, g, G);
            __syncthreads();
        }
    }
}

int pow2_floor(int x) { int p = 1; while (2 * p <= x) p *= 2; return p; }

// lanes per team, units per step and sweeps in flight for a stage; G = 0: not worth a team
TeamGeom geometry(int mode, int F, int T, int L, int Q, int LA, int n_thr) {
    TeamGeom tg{0, 0, 0};
    const int sk = L + 1, D = Q * sk;
    if (mode == MODE_ONLINE) {
        // sweeps in flight: a slot's next sweep starts D nsl steps after its current one, which lasts at most F + sk LA steps
        tg.nsl = (F - 1 + sk * LA) / D + 1;
        tg.nunits = tg.nsl * (LA + 1);
    } else {
        int lpi = (F - 1) / sk + 1;
        if (lpi > T) lpi = T;
        // more sweeps in flight mean fewer steps in all but thinner teams: keep at least 8 lanes per bin
        int group = 1024 / (8 * lpi);
        if (group < 1) group = 1;
        if (group > n_thr) group = n_thr;
        tg.nsl = group;
        tg.nunits = group * lpi;
    }
    if (tg.nunits <= 0 || tg.nunits > 512) return TeamGeom{0, 0, 0};
    int G = pow2_floor(1024 / tg.nunits);
    if (G > 64) G = 64;
    const char *ev = getenv("LWS_TEAM_LANES");   // comparison runs: at most this many lanes per bin (1: the generic engine's order of terms)
    if (ev && atoi(ev) >= 1 && atoi(ev) < G) G = pow2_floor(atoi(ev));
    tg.G = G;
    return tg;
}

// the online ring: rows, what goes into LDS, bytes; bytes = 0: does not fit (the state stays in memory: k_team_online)
template <typename real>
RingGeom ring_geometry(const TeamGeom &tg, int F, int L, int Q, int Qp, int LA, int n_thr) {
    RingGeom rg{};
    const int per = n_thr + 1, Np = F + 2 * L, RQ = Q * (L + 1), NT = L + (Q - 1) * (2 * L + 1);
    rg.DM = (tg.nsl + per - 2) / per;
    rg.NWR = Q + LA + rg.DM;
    rg.NWA = LA + rg.DM + 2;
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t csz = 2 * sizeof(real);
    size_t off = up16((size_t)NT * sizeof(int));
    rg.off_ring = (unsigned)off; off = up16(off + (size_t)(rg.NWR + 1) * Np * csz);   // (+ the row of zeros)
    rg.off_amp = (unsigned)off; off = up16(off + (size_t)rg.NWA * Np * sizeof(real));
    const size_t wbytes = up16((size_t)3 * Qp * RQ * csz);
    const size_t cap = 160 * 1024;
    if (off > cap) return rg;                                   // (bytes = 0)
    rg.wl = (off + wbytes <= cap) ? 1 : 0;
    if (rg.wl) {
        rg.off_w = (unsigned)off; off = up16(off + (size_t)3 * Qp * RQ * csz);
    }
    rg.bytes = (unsigned)off;
    rg.fits = 1;                                                // (... whether or not LWS_TEAM_NO_RING sends the call to the other kernel)
    const char *ev = getenv("LWS_TEAM_NO_RING");                // comparison runs: the state stays in memory
    if (ev && atoi(ev)) rg.bytes = 0;
    { const char *ep = getenv("LWS_TEAM_DBG_POISON"); rg.poison = (ep && atoi(ep) && rg.bytes && rg.bytes + 16384 <= cap) ? 1 : 0; }
    return rg;
}

// fp64 plans: the order-exact online kernel unless LWS_TEAM_FP64=1 asks for the re-associating ones; fp32 plans: with LWS_TEAM_ORDERED=1
bool team_ordered(bool fp64) {
    const char *eo = getenv("LWS_TEAM_ORDERED"), *ef = getenv("LWS_TEAM_FP64");
    return fp64 ? !(ef && atoi(ef)) : (eo && atoi(eo));
}

}  // namespace

bool team_online_is_ordered(bool fp64) { return team_ordered(fp64); }
bool team_ordered_fits(int F, int T, int L, int Q, int LA, int n_thr, bool fp64) {
    const TeamGeom tg = geometry(MODE_ONLINE, F, T, L, Q, LA, n_thr);
    if (tg.G < 1) return false;
    const int NT = L + (Q - 1) * (2 * L + 1), NTP = (NT + 1) | 1;
    const size_t csz = fp64 ? 16 : 8;
    return (size_t)NT * sizeof(Term) + (size_t)tg.nunits * NTP * csz + (size_t)tg.nunits * 32 + 64 <= 160 * 1024;
}

bool team_supports(int mode, int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    if (mode != MODE_ONLINE && mode != MODE_NOFUTURE) return false;
    if (F < 2 || T < 1 || L < 1 || Q < 2 || Q > 255 || Qp < 1 || n_thr < 1) return false;
    const int NT = L + (Q - 1) * (2 * L + 1);
    if ((size_t)NT * sizeof(Term) > 60 * 1024) return false;
    return geometry(mode, F, T, L, Q, LA, n_thr).G >= (getenv("LWS_TEAM_LANES") ? 1 : 2);
}

bool team_online_in_lds(bool fp64, int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    if (!team_supports(MODE_ONLINE, F, T, L, Q, Qp, LA, n_thr)) return false;
    const TeamGeom tg = geometry(MODE_ONLINE, F, T, L, Q, LA, n_thr);
    if (tg.G < 8) return false;
    return (fp64 ? ring_geometry<double>(tg, F, L, Q, Qp, LA, n_thr) : ring_geometry<float>(tg, F, L, Q, Qp, LA, n_thr)).bytes != 0;
}

int team_lanes(int mode, int F, int T, int L, int Q, int LA, int n_thr) { return geometry(mode, F, T, L, Q, LA, n_thr).G; }

template <typename real>
hipError_t launch_team(const GenericArgs<real> &a, int B, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    TeamGeom tg = geometry(a.mode, a.F, a.T, a.L, a.Q, a.LA, a.n_thr);
    if (tg.G < 1) return hipErrorInvalidValue;
    int threads = ((tg.nunits * tg.G + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    const int NT = a.L + (a.Q - 1) * (2 * a.L + 1);
    const size_t lds = (size_t)NT * sizeof(Term);
    if (a.mode == MODE_ONLINE && team_ordered(sizeof(real) == 8)) {
        // order-exact variant (fp64 plans by default; LWS_TEAM_ORDERED=1: fp32 plans too)
        const int NTP = (NT + 1) | 1;                                  // (odd row length: the bins' chains read different banks)
        auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
        const size_t off_inc = up16((size_t)NT * sizeof(Term)), off_unit = up16(off_inc + (size_t)tg.nunits * NTP * 2 * sizeof(real));
        const size_t bytes = off_unit + (size_t)tg.nunits * sizeof(OrdUnit<real>);
        if (bytes > 160 * 1024) return hipErrorInvalidValue;           // (team_supports said no)
        static std::atomic<unsigned long long> done{0};
        int dev;
        if (attr_needed(done, &dev)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_team_online_ordered<real>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done(done, dev);
        }
        hipLaunchKernelGGL(k_team_online_ordered<real>, dim3(B), dim3(threads), bytes, stream, a, tg, (unsigned)off_inc, (unsigned)off_unit, NTP);
        return hipGetLastError();
    }
    if (a.mode == MODE_ONLINE) {
        const RingGeom rg = ring_geometry<real>(tg, a.F, a.L, a.Q, a.Qp, a.LA, a.n_thr);
        // Fewer, fatter teams: what a step costs beside the taps -- schedule bookkeeping, the team sum, the barrier -- is per WAVE, so
        // the ring kernel wants the smallest team whose lanes keep all their terms' placement in registers (8 terms in fp32, 4 in
        // fp64), not the largest the workgroup has room for: lws(1024,256,L=8) 366 -> 298 ms with 8 lanes a bin instead of 16.  (The
        // kernel that leaves the state in memory runs with the same teams when it stands in for the ring kernel: same bits.)
        int nch = 2;
        if (rg.fits && !getenv("LWS_TEAM_LANES")) {
            auto smallest = [&](int nc) { int gp = 1; while (gp * nc < NT) gp *= 2; return gp; };
            int gp = smallest(sizeof(real) == 8 ? 4 : 8);
            if (sizeof(real) == 4 && !getenv("LWS_TEAM_NO_NCH3")) {            // (fp32: three chunks in registers if that halves the team
                const int g3 = smallest(12);                                      //  and the workgroup stays within 512 threads)
                if (g3 < gp && g3 <= tg.G && tg.nunits * g3 <= 512) { gp = g3; nch = 3; }
            }
            if (gp < tg.G) {
                tg.G = gp;
                threads = ((tg.nunits * tg.G + 63) / 64) * 64;
            } else nch = 2;
        }
        { const char *e3 = getenv("LWS_TEAM_NCH3"); if (e3 && atoi(e3) && rg.fits && threads <= 512) nch = 3; }   // (tests: the three-chunk kernel whatever the team)
        if (rg.bytes) {
            auto launch = [&](auto kern) {
                static std::atomic<unsigned long long> done{0};
                int dev;
                if (attr_needed(done, &dev)) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return e;
                    attr_done(done, dev);
                }
                hipLaunchKernelGGL(kern, dim3(B), dim3(threads), rg.bytes + (rg.poison ? 16384 : 0), stream, a, tg, rg);
                return hipGetLastError();
            };
            if (nch == 3) return rg.wl ? launch(&k_team_online_ring<real, true, 3>) : launch(&k_team_online_ring<real, false, 3>);
            return rg.wl ? launch(&k_team_online_ring<real, true, 2>) : launch(&k_team_online_ring<real, false, 2>);
        }
        hipLaunchKernelGGL(k_team_online<real>, dim3(B), dim3(threads), lds, stream, a, tg);
    } else hipLaunchKernelGGL(k_team_sweeps<real>, dim3(B), dim3(threads), lds, stream, a, tg);
    return hipGetLastError();
}

template hipError_t launch_team<float>(const GenericArgs<float> &, int, hipStream_t);
template hipError_t launch_team<double>(const GenericArgs<double> &, int, hipStream_t);

}  // namespace lws
