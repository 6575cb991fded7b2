// lws_online.hip -- LDS-resident engine for the online driver TF_RTISI_LA (lwslib.cpp:1424-1492), fp32.
//
// The online algorithm touches a short moving window of frames: the sweeps belonging to the newest frame m update
// frames m-LA .. m and read Q-1 frames further back, and at lag D = Q(L+1) between consecutive sweeps (the order-exact
// schedule of lws_generic.hip: bin (sweep s, frame rho, bin c) runs at step t = D*s + (L+1)*rho + c) the sweeps in
// flight span a handful of newest frames.  So one workgroup keeps a ring of NW extended frames (state + target
// magnitude) of its spectrogram in LDS, every tap is an LDS read, and HBM sees each frame once on the way in and once
// on the way out (written back when it leaves the ring), instead of ~150 uncoalesced L2 gathers per bin.
//
// Work layout.  Sweep s is owned by slot s mod NSW for its whole life; a slot has (LA+1) frame positions, a frame
// position has Q lanes, lane r sums the taps of frames rho-r / rho+r (lane 0: the centre frame), the Q partial sums
// are combined with cross-lane shuffles and lane 0 re-projects and writes.  One barrier per step.
//
// Same arithmetic per tap as the generic engine (grouped pairs, zero weights where the reference's flag is off); only
// the summation order across frames differs (per-lane partial sums), which is rounding-level in fp32.
#include "lws_common.h"
#include "lws_online.h"

#include <cstdlib>
#include <type_traits>

namespace lws {
namespace {

#ifndef LWS_ON_DBG
#define LWS_ON_DBG 0   // timing experiments only (results invalid): 1 no barrier, 2 no bin updates
#endif
constexpr int NW = 16;  // frames in the LDS ring

struct OnlineArgs {
    float2 *state;       // [B][Tp][Np]
    const float *amp;    // [B][Tp][Np]
    const float *thr;    // [B][n_thr]
    const float2 *w[3];  // W, W_ai, W_af: [Q][Q][L+1], zero where flagged off
    int F, T, n_thr, LA, NSW;
};

__device__ __forceinline__ void pair(float2 &a, float2 w, float2 b, float2 c) {
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}

// production arithmetic: the same grouped form with fused multiply-adds
__device__ __forceinline__ void pair_fma(float2 &a, float2 w, float2 b, float2 c) {
    a.x = fmaf(-w.y, b.y - c.y, fmaf(w.x, b.x + c.x, a.x));
    a.y = fmaf(w.y, b.x - c.x, fmaf(w.x, b.y + c.y, a.y));
}

// One ds_read_b64 (2 LDS cycles per wave, bank = dword address mod 64).  Plain loads get fused by the compiler into
// ds_read2_b64, which the LDS serves at half that rate (MI355X_MICROARCH.md, LDS table); `addr` is a byte offset into
// the dynamic LDS segment, which starts at LDS address 0 (no static __shared__ objects in this kernel).
__device__ __forceinline__ float2 lds_read64(int addr) {
    using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
    const unsigned long long u = *(lds_u64 *)(unsigned)addr;
    return make_float2(__uint_as_float((unsigned)(u & 0xffffffffull)), __uint_as_float((unsigned)(u >> 32)));
}

__device__ __forceinline__ void cmac(float2 &a, float2 w, float2 v) {    // a += w * v
    a.x = fmaf(-w.y, v.y, fmaf(w.x, v.x, a.x));
    a.y = fmaf(w.y, v.x, fmaf(w.x, v.y, a.y));
}
__device__ __forceinline__ void cmacc(float2 &a, float2 w, float2 v) {   // a += conj(w) * v
    a.x = fmaf(w.y, v.y, fmaf(w.x, v.x, a.x));
    a.y = fmaf(-w.y, v.x, fmaf(w.x, v.y, a.y));
}

// sum over the Q adjacent lanes of a bin (Q = 2, 4, 8; groups are aligned), in data-parallel-primitive moves
template <int Q> __device__ __forceinline__ float quad_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>());                    // quad_perm [1,0,3,2]
    if constexpr (Q >= 4) v += dpp(v, std::integral_constant<int, 0x4E>());   // quad_perm [2,3,0,1]
    if constexpr (Q >= 8) v += dpp(v, std::integral_constant<int, 0x141>());  // row_half_mirror: the other quad of the 8
    if constexpr (Q >= 16) v += dpp(v, std::integral_constant<int, 0x140>()); // row_mirror: the other half of the 16
    return v;
}

// SERIAL: verification variant -- lane 0 of a bin sums every tap itself in the generic engine's order, which makes the
// result bit-identical to lws_generic.hip's fp32 online mode (same schedule, same arithmetic); tests use it to pin the
// window / slot logic at sizes where fp32-vs-fp64 comparisons are dominated by the algorithm's own sensitivity.
// H: lanes per frame pair.  1: one lane sums the taps of frames rho-r and rho+r.  2: one lane each -- half the
// instructions per wave and twice the waves, which is what a step (one dependent chain per wave, then a barrier) wants.
template <int Q, int L, bool SERIAL, int H>
__global__ void __launch_bounds__(1024) k_online(OnlineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K1 = L + 1, SK = L + 1, D = Q * SK;
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    const int rps = LA + 1, per = a.n_thr + 1;
    const int nsweeps = T * per;
    float2 *S = reinterpret_cast<float2 *>(smem);                 // [NW][Np]
    float *A = reinterpret_cast<float *>(S + (size_t)NW * Np);      // [NW][Np]
    float2 *W = reinterpret_cast<float2 *>(A + (size_t)NW * Np + ((NW * Np) & 1));   // [3][Q][Q][K1]
    constexpr int WSET = Q * Q * K1 + 1;   // entries per weight set; odd: lanes on different sets hit different banks
    float *thr_s = reinterpret_cast<float *>(W + 3 * WSET);  // [n_thr]
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    float2 *gS = a.state + (size_t)b * Tp * Np;
    const float *gA = a.amp + (size_t)b * Tp * Np;

    // weights; the self weight W[.][0][0] is not part of the sum (update type 2, lws.pyx:363)
    for (int i = tid; i < 3 * Q * Q * K1; i += nthr) {
        const int set = i / (Q * Q * K1), x = i % (Q * Q * K1);
        W[set * WSET + x] = (x % (Q * K1) == 0) ? make_float2(0.f, 0.f) : a.w[set][x];
    }
    // ring slots that have not received a frame yet are read (with zero gain) by lanes whose right-hand frames do not
    // exist yet: they must hold finite numbers
    for (int i = tid; i < NW * Np; i += nthr) S[i] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < a.n_thr; i += nthr) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    // rows 0 .. Q-1 (left edge pads and the first frame) are needed at step 0
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;
    for (int i = tid; i < loaded * Np; i += nthr) { S[i] = gS[i]; A[i] = gA[i]; }

    // this lane: tap group r of frame position j of sweep slot sigma
    const int h = tid % H, r = (tid / H) % Q, j = (tid / (Q * H)) % rps, sigma = tid / (Q * H * rps);
    const bool lane_used = sigma < NSW;
    int s = sigma;
    // per-sweep constants of the lane
    int rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0;
    int lf_base = 0, rt_base = 0, w_base = 0;   // LDS element offsets of frames rho-r / rho+r (column 0) and of W[wset][.][r][0]
    bool valid = false, centre = false, both = false;
    float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
    const int xsgn = (r == 0) ? 1 : -1;
    // H == 2: lane h = 0 takes the terms of the left frame (W lf[-k], conj(W') lf[+k]), lane h = 1 those of the right
    // frame (conj(W) rt[-k], W' rt[+k]); for the centre frame (r = 0) both read it, h = 1 at +k.
    const int side_a = (h == 0) ? -1 : xsgn;            // sign of k for the first sum
    const float sgn_a = (h == 0) ? 1.f : -1.f;          // conj(W) for the right frame
    const float sgn_b = (h == 0) ? -1.f : 1.f;          // conj(W') for the left frame
    float ga = 0.f, gb = 0.f;
    int my_base = 0;
    float thr = 0.f;
    auto setup = [&]() {
        const int m = s / per, q = s - m * per;
        const int first = m - LA > 0 ? m - LA : 0;
        if (q == 0) { valid = (j == 0); rho = m; wset = 1; centre = false; ts = 1; thr = 0.f; }
        else {
            rho = first + j; valid = rho <= m; wset = (rho == m) ? 2 : 0; centre = true;
            ts = m - rho + 1; if (ts > Q) ts = Q;
            thr = thr_s[q - 1];
        }
        valid = valid && lane_used && s < nsweeps;
        tstart = D * s + SK * rho;
        t_done = D * s + SK * m + F - 1;   // last step of the sweep (its newest frame's last bin)
        const int e = rho + Q - 1;
        lf_base = ((e - r) & (NW - 1)) * Np + L;
        rt_base = ((e + r) & (NW - 1)) * Np + L;
        w_base = wset * WSET + r * K1;
        both = (r == 0) ? centre : (r < ts);   // lane 0 pairs the centre frame with itself
        // gains of the four kinds of term (see the bin update).  Lane 0: W[row][0][k] (lf[-k] + conj lf[+k]) if the
        // centre frame takes part, the k = 0 weight slot of the centre frame is zero by construction of the tables.
        g1 = (r == 0) ? (centre ? 1.f : 0.f) : 1.f;
        g2 = both ? 1.f : 0.f;
        g3 = (r != 0 && both) ? 1.f : 0.f;
        g4 = (r != 0) ? 1.f : 0.f;
        ga = (h == 0) ? g1 : g2;
        gb = (h == 0) ? g4 : g3;
        my_base = (h == 0) ? lf_base : rt_base;
    };
    __syncthreads();
    setup();

    const int t_end = D * (nsweeps - 1) + SK * (T - 1) + F;
    int next_need = (loaded - (Q - 1)) * (D * per + SK);   // first step that touches row `loaded`: its frame's first sweep
    for (int t = 0; t < t_end; ++t) {
        const int c = t - tstart;
#if LWS_ON_DBG == 2
        if (false) {
#else
        if (valid && c >= 0 && c < F) {
#endif
            const int e = rho + Q - 1, n = c + L;
            const int row = c % Q, rowneg = (Q - row) % Q;
            const float2 *wa = W + wset * WSET + row * Q * K1;   // centre-frame weights
            float2 acc = make_float2(0.f, 0.f);
            const float2 zero = make_float2(0.f, 0.f);
            if constexpr (SERIAL) {
                if (r == 0) {
                    if (centre) {
                        const float2 *ctr = S + (e & (NW - 1)) * Np + n;
#pragma unroll
                        for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
                    }
#pragma unroll
                    for (int rr = 1; rr < Q; ++rr) {
                        const float2 *lf = S + ((e - rr) & (NW - 1)) * Np + n;
                        const float2 *rt = S + ((e + rr) & (NW - 1)) * Np + n;
                        const float2 *wa_r = W + wset * WSET + (row * Q + rr) * K1;
                        const float2 *wb_r = W + wset * WSET + (rowneg * Q + rr) * K1;
                        const bool two = rr < ts;
                        // a frame to the right that is not usable yet contributes a zero: pair(w, b, 0) == w*b,
                        // pair(w, 0, c) == conj(w)*c, exactly the one-sided forms of lwslib.cpp:1222-1253
                        pair(acc, wa_r[0], lf[0], two ? rt[0] : zero);
#pragma unroll
                        for (int k = 1; k <= L; ++k) {
                            pair(acc, wa_r[k], lf[-k], two ? rt[-k] : zero);
                            pair(acc, wb_r[k], two ? rt[k] : zero, lf[k]);
                        }
                    }
                }
            } else if constexpr (H == 2) {
                const int fb = (my_base + c) * 8;                                                // S starts at LDS byte 0
                const int w_off = (int)((NW * Np) * 12) + w_base * 8;                           // W follows S and A
                const int wa_r = w_off + row * (Q * K1 * 8), wb_r = w_off + rowneg * (Q * K1 * 8);
                float2 va[K1], vb[K1], wA[K1], wB[K1];
#pragma unroll
                for (int k = 0; k <= L; ++k) { va[k] = lds_read64(fb + side_a * 8 * k); wA[k] = lds_read64(wa_r + 8 * k); }
#pragma unroll
                for (int k = 1; k <= L; ++k) { vb[k] = lds_read64(fb + 8 * k); wB[k] = lds_read64(wb_r + 8 * k); }
                __builtin_amdgcn_sched_barrier(0);
                float2 pa = zero, pb = zero;
#pragma unroll
                for (int k = 0; k <= L; ++k) cmac(pa, make_float2(wA[k].x, sgn_a * wA[k].y), va[k]);
#pragma unroll
                for (int k = 1; k <= L; ++k) cmac(pb, make_float2(wB[k].x, sgn_b * wB[k].y), vb[k]);
                acc.x = quad_sum<Q * H>(fmaf(gb, pb.x, ga * pa.x));
                acc.y = quad_sum<Q * H>(fmaf(gb, pb.y, ga * pa.y));
            } else {
                // One instruction stream for all Q lanes of the bin.  Lanes r >= 1: frames rho-r (lf) and rho+r (rt),
                // weights W[row][r][k] for the taps at -k and W[-row][r][k] for the taps at +k.  Lane 0: lf == rt == the
                // centre frame, taps (-k, +k) under W[row][0][k], nothing else.  The four kinds of term
                //   P1 = sum W[row] lf[-k]   P2 = sum conj(W[row]) rt[-k]   P3 = sum W[-row] rt[k]   P4 = sum conj(W[-row]) lf[k]
                // are accumulated separately and switched on or off once per bin by 0/1 gains that are constants of
                // the lane's sweep (frames to the right not usable yet: P2 = P3 = 0, lwslib.cpp:1222-1253), which keeps
                // per-tap selects out of the loop.  A step is one dependent chain per wave, so every LDS read is
                // issued up front and waited for once.
                const int lf = (lf_base + c) * 8, rt = (rt_base + c) * 8;                     // S starts at LDS byte 0
                const int w_off = (int)((NW * Np) * 12) + w_base * 8;                           // W follows S and A
                const int wa_r = w_off + row * (Q * K1 * 8), wb_r = w_off + rowneg * (Q * K1 * 8);
                float2 sl[2 * L + 1], sr[2 * L + 1], wA[K1], wB[K1];
#pragma unroll
                for (int k = -L; k <= L; ++k) sl[k + L] = lds_read64(lf + 8 * k);
                sr[L] = lds_read64(rt);
#pragma unroll
                for (int k = 1; k <= L; ++k) { sr[L - k] = lds_read64(rt + xsgn * 8 * k); sr[L + k] = lds_read64(rt + 8 * k); }   // lane 0 pairs -k with +k
#pragma unroll
                for (int k = 0; k <= L; ++k) { wA[k] = lds_read64(wa_r + 8 * k); wB[k] = lds_read64(wb_r + 8 * k); }
                __builtin_amdgcn_sched_barrier(0);
#if LWS_ON_DBG == 4
#pragma unroll
                for (int k = 1; k <= 2 * L; ++k) { sl[k] = sl[0]; sr[k] = sl[0]; }
#pragma unroll
                for (int k = 1; k <= L; ++k) { wA[k] = wA[0]; wB[k] = wA[0]; }
#endif
                float2 p1 = zero, p2 = zero, p3 = zero, p4 = zero;
                cmac(p1, wA[0], sl[L]);
                cmacc(p2, wA[0], sr[L]);
#pragma unroll
                for (int k = 1; k <= L; ++k) {
                    cmac(p1, wA[k], sl[L - k]);
                    cmacc(p2, wA[k], sr[L - k]);
                    cmac(p3, wB[k], sr[L + k]);
                    cmacc(p4, wB[k], sl[L + k]);
                }
                acc.x = fmaf(g4, p4.x, fmaf(g3, p3.x, fmaf(g2, p2.x, g1 * p1.x)));
                acc.y = fmaf(g4, p4.y, fmaf(g3, p3.y, fmaf(g2, p2.y, g1 * p1.y)));
                acc.x = quad_sum<Q>(acc.x);
                acc.y = quad_sum<Q>(acc.y);
            }
            if (r == 0 && h == 0) {
                const int li = (e & (NW - 1)) * Np + n;
                const float target = A[li];
                if (target > thr) {
                    float2 v;
                    bool nonzero;
#if LWS_ON_DBG == 3
                    if (true) { v = acc; nonzero = true; } else
#endif
                    if constexpr (SERIAL) {
                        const float mag = sqrtf(acc.x * acc.x + acc.y * acc.y);
                        nonzero = mag > 0.f;
                        v = make_float2(acc.x * target / mag, acc.y * target / mag);
                    } else {
                        // target / |acc| as target * rsqrt(|acc|^2) with one Newton step (relative error < 2^-22); sums too
                        // small to square in fp32 are rescaled first so that "|acc| > 0" keeps the reference's meaning
                        float m2 = acc.x * acc.x + acc.y * acc.y;
                        const bool tiny = m2 < 1e-30f;
                        const float ax = tiny ? acc.x * 0x1p60f : acc.x, ay = tiny ? acc.y * 0x1p60f : acc.y;
                        m2 = tiny ? ax * ax + ay * ay : m2;
                        nonzero = m2 > 0.f;
                        float rs = __frsqrt_rn(m2);
                        rs = rs * fmaf(-0.5f * m2 * rs, rs, 1.5f);
                        const float sc = target * rs;
                        v = make_float2(ax * sc, ay * sc);
                    }
                    if (nonzero) {
                        const float2 vc = make_float2(v.x, -v.y);
                        S[li] = v;
                        // Hermitian images in the pad columns (lwslib.cpp:362-367)
                        const int nyq = F + L - 1;
                        if (n >= L + 1 && n < 2 * L + 1) S[li + 2 * (L - n)] = vc;
                        else if (n >= F - 1 && n < nyq) S[li + 2 * (nyq - n)] = vc;
                    }
                }
            }
        }
        if (t >= t_done) { s += NSW; setup(); }
        // bring in the next frame just before the step that first touches it (its own first sweep)
        while (loaded < T + Q - 1 && next_need <= t + 1) {
            const int slot = (loaded & (NW - 1)) * Np;
            const bool evict = loaded >= NW;   // the frame leaving the ring is final: write it back (HBM sees it once)
            for (int i = tid; i < Np; i += nthr) {
                if (evict) gS[(size_t)(loaded - NW) * Np + i] = S[slot + i];
                S[slot + i] = gS[(size_t)loaded * Np + i];
                A[slot + i] = gA[(size_t)loaded * Np + i];
            }
            ++loaded;
            next_need += D * per + SK;
        }
#if LWS_ON_DBG != 1
        __syncthreads();
#endif
    }
    // frames still in the ring
    const int first_row = loaded > NW ? loaded - NW : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e & (NW - 1)) * Np;
        for (int i = tid; i < Np; i += nthr) gS[(size_t)e * Np + i] = S[slot + i];
    }
}

template <int Q, int L, bool SERIAL, int H> hipError_t launch_q(const OnlineArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online<Q, L, SERIAL, H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_online<Q, L, SERIAL, H>), dim3(B), dim3(threads), lds, s, a);
    return hipGetLastError();
}

struct Shape { int NSW, threads, threads2; size_t lds; bool ok; };

Shape shape_of(int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    Shape sh{0, 0, 0, 0, false};
    if (Qp != Q || L != 5 || !(Q == 2 || Q == 4 || Q == 8) || LA < 0 || n_thr < 1 || T < 1) return sh;
    const int SK = L + 1, D = Q * SK, Np = F + 2 * L, per = n_thr + 1;
    sh.NSW = (F - 1 + SK * LA) / D + 2;                       // > sweeps in flight
    sh.threads = ((sh.NSW * (LA + 1) * Q + 63) / 64) * 64;
    sh.threads2 = ((sh.NSW * (LA + 1) * Q * 2 + 63) / 64) * 64;   // two lanes per frame pair
    if (sh.threads > 1024) return sh;
    // Frames alive at once.  The frame loaded at the end of step t (newest frame m_new, (D*per + SK) m_new <= t + 1)
    // replaces the one NW rows below it, and the oldest sweep still running (of frame m_lo, t <= D (per m_lo + per - 1)
    // + SK m_lo + F - 1) reads down to row m_lo - LA:  m_new - m_lo <= (D (per-1) + F) / (D per + SK), and the ring
    // must hold that many frames plus the Q - 1 + LA behind m_lo and the new one.
    const int window = (D * (per - 1) + F) / (D * per + SK) + LA + Q;
    if (window > NW) return sh;
    sh.lds = (size_t)NW * Np * 12 + 8 + (size_t)3 * (Q * Q * (L + 1) + 1) * 8 + (size_t)n_thr * 4;
    if (sh.lds > 160 * 1024) return sh;
    if ((double)D * T * per + (double)SK * T + F > 1.0e9) return sh;   // step counter is an int
    sh.ok = true;
    return sh;
}

}  // namespace

bool online_lds_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update) {
    return update == 2 && shape_of(F, T, L, Q, Qp, LA, n_thr).ok;
}

hipError_t launch_online_lds(const GenericArgs<float> &g, int B, hipStream_t stream) {
    const Shape sh = shape_of(g.F, g.T, g.L, g.Q, g.Qp, g.LA, g.n_thr);
    if (!sh.ok) return hipErrorInvalidValue;
    OnlineArgs a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    for (int i = 0; i < 3; ++i) a.w[i] = g.w[i].w;
    a.F = g.F; a.T = g.T; a.n_thr = g.n_thr; a.LA = g.LA; a.NSW = sh.NSW;
    const char *ev = getenv("LWS_ONLINE_SERIAL_TAPS");   // verification only, see k_online
    if (ev && ev[0] == '1') {
        if (g.Q == 4) return launch_q<4, 5, true, 1>(a, B, sh.threads, sh.lds, stream);
        if (g.Q == 2) return launch_q<2, 5, true, 1>(a, B, sh.threads, sh.lds, stream);
        return launch_q<8, 5, true, 1>(a, B, sh.threads, sh.lds, stream);
    }
    // two lanes per frame pair (half the instructions per wave, twice the waves) measured 9 % slower than one
    // (129 vs 118.6 ms on config 3): kept selectable for re-measurement
    const char *e1 = getenv("LWS_ONLINE_LANES");
    if (sh.threads2 <= 1024 && e1 && e1[0] == '2') {
        if (g.Q == 4) return launch_q<4, 5, false, 2>(a, B, sh.threads2, sh.lds, stream);
        if (g.Q == 2) return launch_q<2, 5, false, 2>(a, B, sh.threads2, sh.lds, stream);
        return launch_q<8, 5, false, 2>(a, B, sh.threads2, sh.lds, stream);
    }
    if (g.Q == 4) return launch_q<4, 5, false, 1>(a, B, sh.threads, sh.lds, stream);
    if (g.Q == 2) return launch_q<2, 5, false, 1>(a, B, sh.threads, sh.lds, stream);
    return launch_q<8, 5, false, 1>(a, B, sh.threads, sh.lds, stream);
}

}  // namespace lws
