// lws_online64.hip -- the online driver TF_RTISI_LA (lwslib.cpp:1424-1492) of an fp64 plan with the frames of the moving window in
// LDS: the reference's own arithmetic type, every sum in the order-exact engine's order (lws_generic.hip: accumulate_v / update_bin),
// so results are BIT-IDENTICAL to generic_fp64 -- which served this stage until round 5 at 2.55 s for config 3's 256 x 500 x 513
// (one bin per step, 76 dependent tap loads per bin from L2).
//
// One workgroup = one spectrogram, on one wave (k_online64, described here) or on two (k_online64p below: the kernel that runs; the
// one-wave kernel stays for comparison, LWS_ONLINE64_ONE_WAVE).  A lane is a (sweep slot, frame position) unit, exactly the units of the fp32
// engine's fourth layout (lws_online.hip: k_online4) on the schedule of its verification variant: two bins per step, frames of a
// sweep SKS steps apart, sweeps DS steps apart (order-exact: every writer of a neighbouring frame or sweep is at least L + 2 bins
// from a lane's taps), 64 units in flight.  A lane sums all the taps of its two bins itself, from LDS, with the full weight tensors
// (no twiddle structure assumed: any summarised tensor, Qp == Q).  A single wave needs no barrier: its LDS operations complete in
// order.  fp64 FMA contraction is off for this file (Makefile), as for the generic engine.
//
// Frames whose window does not fit the LDS with its magnitudes (2048-point frames) keep the state rows there and read the magnitudes from
// memory (AMP_LDS = false).
//
// Entry: online64_supports / launch_online64 (lws_online64.h), called by lws_capi.hip: run_stage for MODE_ONLINE of an fp64 plan.
#include "lws_online64.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace lws {
namespace {

constexpr int L = 5, K1 = L + 1;
constexpr int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2;     // bins / steps between consecutive frames of a sweep

struct Args64 {
    double2 *state;        // [B][Tp][Np]
    const double *amp;     // [B][Tp][Np]
    const double *thr;     // [B][n_thr]
    const double2 *w[3];   // W, W_ai, W_af: [Q][Q][L+1], zero where flagged off (lws_capi.hip: upload_weights)
    int F, T, n_thr, LA, NSW, DS, NWR, NPS;
    int stress;            // test hook (LWS_ONLINE64_STRESS): the two-wave kernel's waves idle for pseudo-random times inside their half-steps
};

// a += w b + conj(w) c, the grouped form of lwslib.cpp:310-311 exactly as lws_generic.hip: pair() writes it
__device__ __forceinline__ void pair(double2 &a, const double2 w, const double2 b, const double2 c) {
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}

__device__ __forceinline__ double2 sel(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }

// (Measured and not kept, round 5: four waves sharing out the PRODUCTS of a bin's tap pairs through LDS, wave 0 adding them up in order --
//  the same bits, the same 440 ms: the counters say one wave issues 1 380 vector instructions a step, 68 % of its time, but spreading the
//  1 000 fp64 operations among them over four SIMDs bought nothing and cost 39 KB of LDS; profiles/r05_pmc_sq_online64.json.)
// AMP_LDS: the target magnitudes of the window's frames in LDS beside the state (frames of up to ~700 bins); false: read from memory, one
// value per bin update, requested before the bin's taps are summed (2048-point frames: the LDS holds the state rows only).
template <int Q, bool AMP_LDS>
__global__ void __launch_bounds__(64) k_online64(Args64 a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, DS = a.DS, NWR = a.NWR, NPS = a.NPS;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), NU = (F + 1) / 2;
    const int rps = LA + 1, per = a.n_thr + 1, nsweeps = T * per;
    double2 *S = reinterpret_cast<double2 *>(smem);                       // [NWR][NPS] (+ 8)
    double *A = reinterpret_cast<double *>(S + (size_t)NWR * NPS + 8);    // [NWR][NPS]
    double2 *W = reinterpret_cast<double2 *>(A + (AMP_LDS ? (((size_t)NWR * NPS + 1) & ~(size_t)1) : 0));      // [3][Q][Q][K1] (16-byte aligned: the stride may be odd)
    double *thr_s = reinterpret_cast<double *>(W + 3 * Q * Q * K1);       // [n_thr]
    const int b = blockIdx.x, lane = threadIdx.x, tid = threadIdx.x;
    constexpr int NTH = 64;
    double2 *gS = a.state + (size_t)b * Tp * Np;
    const double *gA = a.amp + (size_t)b * Tp * Np;

    for (int i = tid; i < 3 * Q * Q * K1; i += NTH) W[i] = a.w[i / (Q * Q * K1)][i % (Q * Q * K1)];
    for (int i = tid; i < NWR * NPS + 8; i += NTH) S[i] = make_double2(0.0, 0.0);   // slots no frame has reached are read with zero weight: finite
    if constexpr (AMP_LDS)
        for (int i = tid; i < NWR * NPS; i += NTH) A[i] = 0.0;
    for (int i = tid; i < a.n_thr; i += NTH) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;          // rows 0 .. Q-1 (left edge pads and the first frame) are needed at step 0
    for (int r0 = 0; r0 < loaded; ++r0)
        for (int i = tid; i < Np; i += NTH) { S[r0 * NPS + i] = gS[(size_t)r0 * Np + i]; if constexpr (AMP_LDS) A[r0 * NPS + i] = gA[(size_t)r0 * Np + i]; }

    const int sigma = lane / rps, j = lane - sigma * rps;
    const bool lane_used = sigma < NSW;
    int s = sigma, rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0, ctb = 0;
    int rowL[Q], rowR[Q];               // ring rows (element offsets) of frames rho - rr / rho + rr, per sweep (no division in the step loop)
    bool valid = false, centre = false;
    double thr = 0.0;
    // sweep s: frame m = s / per, q = s % per.  q == 0: first estimate of frame m from the past (W_ai, threshold 0); q >= 1: iteration
    // q - 1 over frames max(0, m - LA) .. m, the look-ahead frames with W, frame m with W_af (lwslib.cpp:1432-1491)
    auto setup = [&]() {
        const int m = s / per, q = s - m * per;
        const int first = m - LA > 0 ? m - LA : 0;
        if (q == 0) { valid = (j == 0); rho = m; wset = 1; centre = false; ts = 1; thr = 0.0; }
        else {
            rho = first + j; valid = rho <= m; wset = (rho == m) ? 2 : 0; centre = true;
            ts = m - rho + 1; if (ts > Q) ts = Q;
            thr = thr_s[q - 1];
        }
        valid = valid && lane_used && s < nsweeps;
        tstart = DS * s + SKS * rho;
        t_done = DS * s + SKS * m + NU - 1;
        ctb = ((rho + Q - 1) % NWR) * NPS;
#pragma unroll
        for (int rr = 1; rr < Q; ++rr) { rowL[rr] = ((rho + Q - 1 - rr) % NWR) * NPS; rowR[rr] = ((rho + Q - 1 + rr) % NWR) * NPS; }
    };
    setup();

    const int t_end = DS * (nsweeps - 1) + SKS * (T - 1) + NU;
    const int frame_period = DS * per + SKS;
    int next_need = (loaded - (Q - 1)) * frame_period;
    auto load_frames = [&](int t) {      // the next frame a few steps before its first sweep starts; the oldest one goes back to memory
        while (loaded < T + Q - 1 && next_need <= t + 4) {
            const int slot = (loaded % NWR) * NPS;
            const bool evict = loaded >= NWR;
            for (int i = tid; i < Np; i += NTH) {
                if (evict) gS[(size_t)(loaded - NWR) * Np + i] = S[slot + i];
                S[slot + i] = gS[(size_t)loaded * Np + i];
                if constexpr (AMP_LDS) A[slot + i] = gA[(size_t)loaded * Np + i];
            }
            ++loaded;
            next_need += frame_period;
        }
    };
    for (int t = 0; t < t_end; ++t) {
        const int u = t - tstart;
        if (valid && u >= 0 && u < NU) {
            const int c = 2 * u, n = c + L;
            const double2 zero = make_double2(0.0, 0.0);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int cb = c + bb, nb = n + bb;
                if (cb >= F) break;
                const int row = cb % Q, rowneg = (Q - row) % Q;
                const double2 *wa = W + wset * Q * Q * K1 + row * Q * K1;
                double target_mem = 0.0;
                if constexpr (!AMP_LDS) target_mem = gA[(size_t)(rho + Q - 1) * Np + nb];   // (in flight while the taps are summed)
                double2 acc = zero;
                if (centre) {
                    const double2 *ctr = S + ctb + nb;
#pragma unroll
                    for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
                }
#pragma unroll
                for (int rr = 1; rr < Q; ++rr) {
                    const double2 *lf = S + rowL[rr] + nb;
                    const double2 *rt = S + rowR[rr] + nb;
                    const double2 *wa_r = W + wset * Q * Q * K1 + (row * Q + rr) * K1;
                    const double2 *wb_r = W + wset * Q * Q * K1 + (rowneg * Q + rr) * K1;
                    const bool two = rr < ts;
                    // a frame to the right that is not usable yet contributes a zero: pair(w, b, 0) == w b, pair(w, 0, c) == conj(w) c,
                    // the one-sided forms of lwslib.cpp:1222-1253 (lws_generic.hip: mac / macc) bit for bit
                    // (fetched whether usable or not, dropped by a select: `two ? rt[..] : zero` makes the compiler branch around every load)
                    { const double2 rv = rt[0]; pair(acc, wa_r[0], lf[0], sel(two, rv, zero)); }
#pragma unroll
                    for (int k = 1; k <= L; ++k) {
                        const double2 rm = rt[-k], rp = rt[k];
                        pair(acc, wa_r[k], lf[-k], sel(two, rm, zero));
                        pair(acc, wb_r[k], sel(two, rp, zero), lf[k]);
                    }
                }
                const int lj = ctb + nb;
                const double target = AMP_LDS ? A[lj] : target_mem;
                if (target > thr) {
                    const double mag = sqrt(acc.x * acc.x + acc.y * acc.y);
                    if (mag > 0.0) {
                        const double2 v = make_double2(acc.x * target / mag, acc.y * target / mag);
                        const double2 vc = make_double2(v.x, -v.y);
                        S[lj] = v;
                        const int nyq = F + L - 1;     // Hermitian images in the pad columns (lwslib.cpp:362-367)
                        if (nb >= L + 1 && nb < 2 * L + 1) S[lj + 2 * (L - nb)] = vc;
                        else if (nb >= F - 1 && nb < nyq) S[lj + 2 * (nyq - nb)] = vc;
                    }
                }
            }
        }
        if (t >= t_done) { s += NSW; setup(); }
        load_frames(t);
        // (one wave: its LDS operations complete in order, the stores above are visible to the reads of the next step)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    const int first_row = loaded > NWR ? loaded - NWR : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e % NWR) * NPS;
        for (int i = tid; i < Np; i += NTH) gS[(size_t)e * Np + i] = S[slot + i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Two waves per spectrogram (round 5): the wave above issues 1 380 vector instructions a step and that is its time.  The two bins of a
// step depend on each other through the centre frame only (bin c + 1 starts from the new bin c: its k = 1 tap, and near DC / Nyquist the
// Hermitian image of it), so wave 0 takes the even bin of every step, wave 1 the odd one, and each splits its bin in two:
//   P  the PRODUCTS of the neighbour frames' tap pairs, w (b + c) ... of lwslib.cpp:310-311, one (re, im) per pair, kept in registers
//      -- they involve other (sweep, frame) units' frames only, which are final steps before (below);
//   C  the chain: the centre frame's five tap pairs, then the products added IN THE ORDER the one-wave kernel adds them, the
//      re-projection and the stores.
// acc += (m1 - m2) is two roundings whether the difference is formed now or a step earlier: the sums -- and the results -- are the
// one-wave kernel's and the generic engine's, bit for bit.  A step is two half-steps, each closed by a barrier of the two waves:
//      wave 0:  C(bin 2u) . P1(bin 2u+2) | P2(bin 2u+2)
//      wave 1:  P2(bin 2u+1)             | C(bin 2u+1) . P1(bin 2u+3)
// with P1 / P2 the first SPLIT / the other pairs, cut so that the halves take the same time (Q = 4: the nearest frame's eleven pairs are formed
// on the chain instead -- ONCH -- and P1 is empty: fewer products to hold, launch_q).  Reading ahead is legal: a product of bin c + 3
// is formed one and a half steps before the one-wave schedule reads its taps (up to bin c + 8 of the frame one to the left, which that
// frame's even-bin wave wrote in the first half of this step; up to c + 8 of the previous sweep's frames, 2 DS - 8 r >= 8 bins ahead:
// shape64 keeps 2 DS >= 8 Q + 2), and reading later-to-be-overwritten values earlier is always safe.
template <class Fn, int... I> __device__ __forceinline__ void static_for64_impl(Fn &f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class Fn> __device__ __forceinline__ void static_for64(Fn &&f) { static_for64_impl(f, std::make_integer_sequence<int, N>{}); }

template <int Q, bool AMP_LDS, int SPLIT, bool STRESS, int ONCH = 0>
__global__ void __launch_bounds__(128) k_online64p(Args64 a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, DS = a.DS, NWR = a.NWR, NPS = a.NPS;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), NU = (F + 1) / 2;
    const int rps = LA + 1, per = a.n_thr + 1, nsweeps = T * per;
    double2 *S = reinterpret_cast<double2 *>(smem);                       // [NWR][NPS] (+ 8)
    double *A = reinterpret_cast<double *>(S + (size_t)NWR * NPS + 8);    // [NWR][NPS] (AMP_LDS)
    double2 *W = reinterpret_cast<double2 *>(A + (AMP_LDS ? (((size_t)NWR * NPS + 1) & ~(size_t)1) : 0));
    double *thr_s = reinterpret_cast<double *>(W + 3 * Q * Q * K1);
    double2 *Z = reinterpret_cast<double2 *>(thr_s + ((a.n_thr + 2) & ~1)) + L;     // eleven zeros, Z[-L .. L]: what a frame to the right that is not usable yet contributes
    const int b = blockIdx.x, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int par = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // 0: the even bin of a step, 1: the odd one
    constexpr int NTH = 128;
    constexpr int NP = (Q - 1) * (2 * L + 1);                             // neighbour tap pairs of a bin, in the order they are added
    constexpr int NPC = ONCH * (2 * L + 1);                               // the pairs of the first ONCH neighbour frames stay on the chain (fewer products to hold)
    static_assert(SPLIT >= NPC && SPLIT <= NP, "cut of the pair list");
    double2 *gS = a.state + (size_t)b * Tp * Np;
    const double *gA = a.amp + (size_t)b * Tp * Np;

    for (int i = tid; i < 3 * Q * Q * K1; i += NTH) W[i] = a.w[i / (Q * Q * K1)][i % (Q * Q * K1)];
    for (int i = tid; i < NWR * NPS + 8; i += NTH) S[i] = make_double2(0.0, 0.0);
    if constexpr (AMP_LDS)
        for (int i = tid; i < NWR * NPS; i += NTH) A[i] = 0.0;
    for (int i = tid; i < a.n_thr; i += NTH) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    if (tid < 2 * L + 1) Z[tid - L] = make_double2(0.0, 0.0);
    __syncthreads();
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;
    for (int r0 = 0; r0 < loaded; ++r0)
        for (int i = tid; i < Np; i += NTH) { S[r0 * NPS + i] = gS[(size_t)r0 * Np + i]; if constexpr (AMP_LDS) A[r0 * NPS + i] = gA[(size_t)r0 * Np + i]; }
    __syncthreads();

    const int sigma = lane / rps, j = lane - sigma * rps;
    const bool lane_used = sigma < NSW;
    int s = sigma, rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0, ctb = 0;
    int rowL[Q], rowR[Q];
    bool valid = false, centre = false;
    double thr = 0.0;
    auto setup = [&]() {      // as in k_online64
        const int m = s / per, q = s - m * per;
        const int first = m - LA > 0 ? m - LA : 0;
        if (q == 0) { valid = (j == 0); rho = m; wset = 1; centre = false; ts = 1; thr = 0.0; }
        else {
            rho = first + j; valid = rho <= m; wset = (rho == m) ? 2 : 0; centre = true;
            ts = m - rho + 1; if (ts > Q) ts = Q;
            thr = thr_s[q - 1];
        }
        valid = valid && lane_used && s < nsweeps;
        tstart = DS * s + SKS * rho;
        t_done = DS * s + SKS * m + NU - 1;
        ctb = ((rho + Q - 1) % NWR) * NPS;
#pragma unroll
        for (int rr = 1; rr < Q; ++rr) { rowL[rr] = ((rho + Q - 1 - rr) % NWR) * NPS; rowR[rr] = ((rho + Q - 1 + rr) % NWR) * NPS; }
    };
    setup();

    const int t_end = DS * (nsweeps - 1) + SKS * (T - 1) + NU;
    const int frame_period = DS * per + SKS;
    int next_need = (loaded - (Q - 1)) * frame_period;
    auto load_frames = [&](int t) {      // wave 0 moves the rows, in its second half-step (the slot it overwrites is a frame no unit reads any more); wave 1 keeps count
        while (loaded < T + Q - 1 && next_need <= t + 4) {
            if (par == 0) {
                const int slot = (loaded % NWR) * NPS;
                const bool evict = loaded >= NWR;
                for (int i = lane; i < Np; i += 64) {
                    if (evict) gS[(size_t)(loaded - NWR) * Np + i] = S[slot + i];
                    S[slot + i] = gS[(size_t)loaded * Np + i];
                    if constexpr (AMP_LDS) A[slot + i] = gA[(size_t)loaded * Np + i];
                }
            }
            ++loaded;
            next_need += frame_period;
        }
    };
    // the bin of this wave in step t of the unit's current sweep: its index, or -1
    auto my_bin = [&](int t) {
        const int u = t - tstart, cb = 2 * u + par;
        return (valid && u >= 0 && u < NU && cb < F) ? cb : -1;
    };
    double px[NP > 0 ? NP : 1], py[NP > 0 ? NP : 1];      // products of the bin whose chain comes next
    double target = 0.0;
    const double2 zero = make_double2(0.0, 0.0);
    // product I of the bin at column nb (weight rows row / rowneg): (re, im) of w (b + c) ... as pair() forms it
    auto product = [&](auto ic, int nb, int row, int rowneg, double &ox, double &oy) {
        constexpr int I = decltype(ic)::value, rr = 1 + I / (2 * L + 1), jj = I % (2 * L + 1), k = (jj + 1) / 2;
        const double2 *lf = S + rowL[rr] + nb;
        // (a frame to the right that is not usable yet contributes zeros: read from the zero block -- one select on the address per frame
        //  instead of four on every value; pair(w, b, 0) == w b, pair(w, 0, c) == conj(w) c, the one-sided forms of lwslib.cpp:1222-1253)
        const double2 *rt = rr < ts ? S + rowR[rr] + nb : Z;
        const double2 *wa_r = W + wset * Q * Q * K1 + (row * Q + rr) * K1;
        const double2 *wb_r = W + wset * Q * Q * K1 + (rowneg * Q + rr) * K1;
        double2 w, bb, cc;
        if constexpr (jj == 0) { w = wa_r[0]; bb = lf[0]; cc = rt[0]; }
        else if constexpr (jj & 1) { w = wa_r[k]; bb = lf[-k]; cc = rt[-k]; }
        else { w = wb_r[k]; bb = rt[k]; cc = lf[k]; }
        ox = w.x * (bb.x + cc.x) - w.y * (bb.y - cc.y);
        oy = w.x * (bb.y + cc.y) + w.y * (bb.x - cc.x);
    };
    auto products = [&](auto lo_c, auto hi_c, int t) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int cb = my_bin(t);
        if (cb < 0) return;
        const int nb = cb + L, row = cb % Q, rowneg = (Q - row) % Q;
        if constexpr (LO == NPC) target = AMP_LDS ? A[ctb + nb] : gA[(size_t)(rho + Q - 1) * Np + nb];
        static_for64<HI - LO>([&](auto ic) {
            constexpr int I = LO + decltype(ic)::value;
            product(std::integral_constant<int, I>{}, nb, row, rowneg, px[I], py[I]);
        });
    };
    auto chain = [&](int t) {
        const int cb = my_bin(t);
        if (cb < 0) return;
        const int nb = cb + L, row = cb % Q;
        double2 acc = zero;
        if (centre) {
            const double2 *wa = W + wset * Q * Q * K1 + row * Q * K1;
            const double2 *ctr = S + ctb + nb;
#pragma unroll
            for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
        }
        static_for64<NPC>([&](auto ic) {      // the nearest frames' pairs, formed here
            double tx, ty;
            product(ic, nb, row, (Q - row) % Q, tx, ty);
            acc.x += tx; acc.y += ty;
        });
        static_for64<NP - NPC>([&](auto ic) { constexpr int I = NPC + decltype(ic)::value; acc.x += px[I]; acc.y += py[I]; });
        const int lj = ctb + nb;
        if (target > thr) {
            const double mag = sqrt(acc.x * acc.x + acc.y * acc.y);
            if (mag > 0.0) {
                const double2 v = make_double2(acc.x * target / mag, acc.y * target / mag);
                const double2 vc = make_double2(v.x, -v.y);
                S[lj] = v;
                const int nyq = F + L - 1;
                if (nb >= L + 1 && nb < 2 * L + 1) S[lj + 2 * (L - nb)] = vc;
                else if (nb >= F - 1 && nb < nyq) S[lj + 2 * (nyq - nb)] = vc;
            }
        }
    };
    auto advance = [&](int t) { if (t >= t_done) { s += NSW; setup(); } };     // the unit's state becomes that of step t + 1
#define O64_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    using c0 = std::integral_constant<int, NPC>;
    using cs = std::integral_constant<int, SPLIT>;
    using cn = std::integral_constant<int, NP>;
    if (par == 0) { products(c0{}, cn{}, 0); }
    else { products(c0{}, cs{}, 0); }
    // (test hook: what a half-step reads must not depend on how far the other wave has got inside it -- the waves idle for
    //  pseudo-random times before and between their parts, and the results must stay the same bits)
    auto jitter = [&](int t, int where) {
        if constexpr (STRESS) {
            const unsigned h = ((unsigned)(t * 4 + where) * 2654435761u + (unsigned)a.stress * 40503u + (unsigned)par * 0x9e3779b9u) >> 24;
            if (h & 1) for (unsigned q = 0; q < ((h >> 1) & 15u); ++q) __builtin_amdgcn_s_sleep(16);
        }
    };
    for (int t = 0; t < t_end; ++t) {
        if (par == 0) {
            jitter(t, 0);
            chain(t);
            advance(t);
            jitter(t, 1);
            products(c0{}, cs{}, t + 1);
            O64_BARRIER();
            jitter(t, 2);
            products(cs{}, cn{}, t + 1);
            load_frames(t);
            O64_BARRIER();
        } else {
            jitter(t, 0);
            products(cs{}, cn{}, t);
            O64_BARRIER();
            jitter(t, 2);
            chain(t);
            advance(t);
            jitter(t, 3);
            products(c0{}, cs{}, t + 1);
            load_frames(t);
            O64_BARRIER();
        }
    }
#undef O64_BARRIER
    __syncthreads();
    const int first_row = loaded > NWR ? loaded - NWR : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e % NWR) * NPS;
        for (int i = tid; i < Np; i += NTH) gS[(size_t)e * Np + i] = S[slot + i];
    }
}

struct Shape64 { int NSW, DS, NWR, NPS; size_t lds; bool ok, amp_lds; };
// The schedule of lws_online.hip: shape4_try for its verification variant (even lag), sized for fp64 rows.
Shape64 shape64(int F, int T, int Lu, int Q, int Qp, int LA, int n_thr) {
    Shape64 r{0, 0, 0, 0, 0, false, true};
    if (Qp != Q || Lu != L || LA < 0 || LA > 63 || n_thr < 1 || T < 1 || !(Q == 2 || Q == 3 || Q == 4 || Q == 8)) return r;
    const int Np = F + 2 * L, per = n_thr + 1, NU = (F + 1) / 2;
    if (F - 1 < 2 * (L + 3)) return r;
    r.NSW = 64 / (LA + 1);
    int DS = (SKB * (Q - 1) + L + 3) / 2;
    if (2 * DS < SKB * Q + 2) DS = (SKB * Q + 3) / 2;
    const int need = (SKS * LA + NU + 2 + r.NSW - 1) / r.NSW;      // a slot is free again when its sweep is over
    if (DS < need) DS = need;
    DS += DS & 1;
    // Row stride.  A lane's tap (frame j' rows from its unit's, 2 DS sigma columns from its neighbour sweep's) is a 16-byte double2: a
    // wave's ds_read_b128 is served in four groups of 16 lanes, conflict-free when the 16 addresses of a group fall into 16 different
    // 16-byte slots of a 256-byte line (MI355X_MICROARCH, LDS).  With the rows Np (+1) columns apart the sweeps' 64-byte steps and the
    // frames' 192-byte steps left four slots for sixteen lanes -- every read took four times its cycles, and a step is 2 x 228 of them: the stride is chosen, among Np .. Np + 15, for the fewest cycles per read (439 -> 419 ms for config 3's stage).
    {
        static const int GRP[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                       {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
        const int rps = LA + 1;
        int best = 0, best_cost = 1 << 30;
        for (int pad = 0; pad < 16; ++pad) {
            int cost = 0;
            for (int g = 0; g < 4; ++g) {
                int cnt[16] = {0}, mx = 0;
                for (int i = 0; i < 16; ++i) {
                    const int lane = GRP[g][i], sg = lane / rps, jj = lane % rps;
                    const long addr = ((long)jj * (Np + pad) + 2L * DS * sg) * 16;
                    mx = std::max(mx, ++cnt[(addr >> 4) & 15]);
                }
                cost += mx;
            }
            if (cost < best_cost) { best_cost = cost; best = pad; }
        }
        r.NPS = Np + best;
    }
    auto window_of = [&](int ds) { return (ds * (per - 1) + NU + 3) / (ds * per + SKS) + LA + Q; };   // frames alive at once
    // the window's state rows and magnitudes in LDS; if that does not fit at the schedule's own pace, the state rows alone (the
    // magnitudes then come from memory, k_online64<.., false>); only then a slower schedule (fewer frames alive at once)
    const int ds0 = DS;
    int nwr_max = 0;
    for (int pass = 0; pass < 3; ++pass) {
        r.amp_lds = pass == 0;
        auto lds_try = [&](int nwr) {
            return ((size_t)nwr * r.NPS + 8) * 16 + (r.amp_lds ? (size_t)nwr * r.NPS * 8 + 8 : 0) + (size_t)3 * Q * Q * K1 * 16 + (size_t)(n_thr + 2) * 8 + 64 + 256;
        };
        nwr_max = 16;
        while (nwr_max > 0 && lds_try(nwr_max) > 160 * 1024) --nwr_max;
        DS = ds0;
        if (pass == 2)
            while (window_of(DS) > nwr_max && DS < 16 * ds0) DS += 2;
        if (window_of(DS) <= nwr_max) break;
    }
    if (window_of(DS) > nwr_max) return r;
    auto lds_of = [&](int nwr) {
        return ((size_t)nwr * r.NPS + 8) * 16 + (r.amp_lds ? (size_t)nwr * r.NPS * 8 + 8 : 0) + (size_t)3 * Q * Q * K1 * 16 + (size_t)(n_thr + 2) * 8 + 64 + 256;
    };
    r.DS = DS;
    const int window = window_of(DS);
    r.NWR = window + 1 <= nwr_max ? window + 1 : window;
    r.lds = lds_of(r.NWR);
    if ((double)DS * T * per + (double)SKS * T + NU > 1.0e9) return r;
    r.ok = true;
    return r;
}

template <int Q, bool AMP_LDS> hipError_t launch_qa(const Args64 &a, int B, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online64<Q, AMP_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_online64<Q, AMP_LDS>), dim3(B), dim3(64), lds, s, a);
    return hipGetLastError();
}
template <int Q, bool AMP_LDS, int SPLIT, bool STRESS, int ONCH = 0> hipError_t launch_qps(const Args64 &a, int B, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online64p<Q, AMP_LDS, SPLIT, STRESS, ONCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_online64p<Q, AMP_LDS, SPLIT, STRESS, ONCH>), dim3(B), dim3(128), lds, s, a);
    return hipGetLastError();
}
template <int Q, bool AMP_LDS, int SPLIT, int ONCH> hipError_t launch_qp(const Args64 &a, int B, size_t lds, hipStream_t s) {
    return a.stress ? launch_qps<Q, AMP_LDS, SPLIT, true, ONCH>(a, B, lds, s) : launch_qps<Q, AMP_LDS, SPLIT, false, ONCH>(a, B, lds, s);
}
// How a bin's (Q - 1)(2L + 1) neighbour pairs are shared out.  The chain (the centre frame's pairs, the additions, the re-projection) costs
// about twelve pairs' products, and every product held across a barrier is two registers: with all 33 pairs of Q = 4 held (cut after 8)
// the kernel needs 256 + 92 registers and a tenth of its instructions are moves to and from the AGPRs -- 373 ms; with the nearest frame's
// eleven pairs formed on the chain and the other 22 in the opposite half-step (no cut) the halves balance and nothing spills: 331 ms
// (cuts after 13 / 15 / 17: 343 / 357 / 366).
// Q = 8 (77 pairs): three frames' pairs on the chain, 44 products held (256 + 200 registers): 980 -> 597 ms for 256 x 250 x 513; four / 33: 650.
template <int Q> constexpr int onch_of() { return Q == 4 ? 1 : (Q == 8 ? 3 : 0); }
template <int Q> constexpr int split_of() { return Q == 4 ? 11 : (Q == 8 ? 33 : ((Q - 1) * (2 * L + 1) > 16 ? ((Q - 1) * (2 * L + 1) - 16) / 2 : 0)); }
template <int Q> hipError_t launch_q(const Args64 &a, int B, size_t lds, bool amp_lds, hipStream_t s) {
    const bool one_wave = getenv("LWS_ONLINE64_ONE_WAVE") != nullptr;      // the one-wave kernel, for comparison (read on every launch: a test sets it between calls)
    {
        if (!one_wave) return amp_lds ? launch_qp<Q, true, split_of<Q>(), onch_of<Q>()>(a, B, lds, s) : launch_qp<Q, false, split_of<Q>(), onch_of<Q>()>(a, B, lds, s);
    }
    return amp_lds ? launch_qa<Q, true>(a, B, lds, s) : launch_qa<Q, false>(a, B, lds, s);
}

}  // namespace

const char *online64_name() { return getenv("LWS_ONLINE64_ONE_WAVE") != nullptr ? "online_lds_fp64_1w" : "online_lds_fp64"; }

bool online64_supports(int F, int T, int Lplan, int Q, int Qp, int LA, int n_thr, int update) {
    return update == 2 && shape64(F, T, Lplan, Q, Qp, LA, n_thr).ok;
}

hipError_t launch_online64(const GenericArgs<double> &g, int B, hipStream_t stream) {
    const Shape64 sh = shape64(g.F, g.T, g.L, g.Q, g.Qp, g.LA, g.n_thr);
    if (!sh.ok || g.update != 2 || g.mode != MODE_ONLINE) return hipErrorInvalidValue;
    if (B <= 0) return hipSuccess;
    Args64 a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    for (int i = 0; i < 3; ++i) a.w[i] = g.w[i].w;
    a.F = g.F; a.T = g.T; a.n_thr = g.n_thr; a.LA = g.LA; a.NSW = sh.NSW; a.DS = sh.DS; a.NWR = sh.NWR; a.NPS = sh.NPS;
    { const char *es = getenv("LWS_ONLINE64_STRESS"); a.stress = es ? atoi(es) : 0; }
    switch (g.Q) {
    case 2: return launch_q<2>(a, B, sh.lds, sh.amp_lds, stream);
    case 3: return launch_q<3>(a, B, sh.lds, sh.amp_lds, stream);
    case 4: return launch_q<4>(a, B, sh.lds, sh.amp_lds, stream);
    default: return launch_q<8>(a, B, sh.lds, sh.amp_lds, stream);
    }
}

}  // namespace lws
