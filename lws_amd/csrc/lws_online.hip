// lws_online.hip -- LDS-resident engine for the online driver TF_RTISI_LA (lwslib.cpp:1424-1492), fp32.
//
// The online algorithm touches a short moving window of frames: the sweeps belonging to the newest frame m update
// frames m-LA .. m and read Q-1 frames further back.  One workgroup keeps a ring of NW extended frames (state + target
// magnitude) of its spectrogram in LDS, so HBM sees each frame once on the way in and once on the way out (written back
// when it leaves the ring).
//
// Schedule.  As everywhere in this library the reference's sequential order is kept by a skewed schedule: sweep s
// (one Asym_UpdatePhase* call of TF_RTISI_LA, in call order) works on frame rho at bins (c, c+1) -- a PAIR per step --
// at step
//        t = DS*s + SKS*rho + c/2 ,     SKB = 2 SKS >= L + 2 bins between consecutive frames,
//                                       DB  = 2 DS  >  SKB (Q-1) + L + 1 bins between consecutive sweeps,
// so that every "new" neighbour (rho-r, c+k) was written in an earlier step and every "old" one (rho+r, c+k) by the
// previous sweep, exactly what the sequential loops read.  (Q = 4, L = 5: SKB = 8, DB = 32, the numbers of the batch
// kernel.)  Two bins per step halve the number of steps, and a step is what costs here: it is one dependent chain
// -- sums, cross-lane reduction, re-projection -- followed by a barrier.
//
// Work layout.  Sweep s is owned by slot s mod NSW for its whole life; a slot has (LA+1) frame positions, a frame
// position has 2Q lanes: lane (r, 0) sums the taps of frame rho-r, lane (r, 1) those of frame rho+r (lane (0, 0): the
// centre frame; lane (0, 1) idles), the 2Q partial sums are combined with data-parallel-primitive moves and lane (0, 0)
// re-projects and writes.  Many light waves rather than few heavy ones: a step is latency, and the waves of a SIMD hide
// each other's (4 waves of 350 instructions per step: 2400 clocks; 5: 2850 -- the fifth cost next to nothing).
//
// Taps live in REGISTERS.  A lane marches along its two frames, so the 2L+2 columns its pair of bins needs from each
// are a window that slides by two columns per step: two LDS reads per frame and step instead of 2(2L+1).  That is
// exact because no other sweep writes inside a lane's window while it holds it: writers of neighbouring frames and
// sweeps are SKB / DB - SKB(Q-1) >= L + 2 bins away at all times (they all advance two bins per step); only the
// centre-frame lane sees its own outputs (and the Hermitian images they imply, lwslib.cpp:362-367) appear inside its
// window, and patches them in.  Weights live in registers too: with summarised weights of create_weights' structure,
// W[row][r][k] = W[0][r][k] exp(2 pi j row r / Q) (verified on the host for every tensor of the plan), a lane needs the
// L+1 base weights of its r and one twiddle per bin.
//
// Same arithmetic per tap as the generic engine; only the summation order differs (per-lane partial sums), which is
// rounding-level in fp32.  SERIAL: verification variant -- lane 0 of a bin sums every tap itself, from LDS, with the
// full weight tensor and in the generic engine's order, which makes the result bit-identical to lws_generic.hip's fp32
// online mode; tests use it to pin the schedule, frame window and slot logic at sizes where fp32-vs-fp64 comparisons
// are dominated by the algorithm's own sensitivity.
#include "lws_common.h"
#include "lws_online.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace lws {
#ifdef LWS_LAB   // tools/online_budget.hip: per-wave phase stamps of k_online3 / k_online4 (block 0), clocks summed over the steps.
// s_memtime returns through the scalar-memory counter, so reading a stamp drains the wave's LDS operations too: level 1 puts
// stamps only where the wave is about to wait for everything in flight anyway; level 2 (-DLWS_LAB=2) adds one in front of every
// barrier (turning the projection wave's counted wait into a full one: who waits for whom, at the price of a longer step).
#define LAB_N 256
__device__ unsigned long long g_lab[LAB_N];
__device__ __forceinline__ unsigned long long lab_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define LAB(...) __VA_ARGS__
#if LWS_LAB >= 2
#define LAB2(...) __VA_ARGS__
#else
#define LAB2(...)
#endif
#else
#define LAB(...)
#define LAB2(...)
#endif
namespace {

constexpr int NW = 16;  // frames in the LDS ring

template <int... Is, typename F_>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F_ &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N_, typename F_> __device__ __forceinline__ void static_for(F_ &&f) {
    static_for_impl(std::make_integer_sequence<int, N_>{}, static_cast<F_ &&>(f));
}

struct OnlineArgs {
    float2 *state;       // [B][Tp][Np]
    const float *amp;    // [B][Tp][Np]
    const float *thr;    // [B][n_thr]
    const float2 *w[3];  // W, W_ai, W_af: [Q][Q][L+1], zero where flagged off
    float2 tw[8];        // exp(2 pi j q / Q), q < Q
    int F, T, n_thr, LA, NSW;
    int DS;              // steps between consecutive sweeps (>= the order-exact minimum, see shape_of)
    int NWR, NPS;        // k_online4: frames in its LDS ring, row stride (elements, even)
    int Lu;              // k_online4: the caller's stencil half-width (<= the kernel's L): its buffers have 2 Lu pad columns and
                         // its weight tensors Lu + 1 columns; the taps it does not have carry weight zero in the kernel's table
    // k_online4<..., TWT>: twiddles exp(2 pi j bin r s / PT) that are not the eighth turns of Q in {2,4,8} -- Q = 3, and the general
    // weights of a hop that does not divide the frame (Asym_UpdatePhasefractionalQ, lwslib.cpp:1276-1421) -- from a table
    const float2 *twt;   // device: [PT + 3][TQ], TQ = 4 (Q <= 4) or 8: row p = tau_0 .. tau_{TQ-1} of bin p (periodic: rows PT .. PT + 2 repeat rows 0 .. 2)
    int PT;              // period of the twiddles in bins
};

__device__ __forceinline__ void pair(float2 &a, float2 w, float2 b, float2 c) {   // the generic engine's grouped form
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}
__device__ __forceinline__ void cmac(float2 &a, float2 w, float2 v) {    // a += w * v
    a.x = fmaf(-w.y, v.y, fmaf(w.x, v.x, a.x));
    a.y = fmaf(w.y, v.x, fmaf(w.x, v.y, a.y));
}
__device__ __forceinline__ void cmacc(float2 &a, float2 w, float2 v) {   // a += conj(w) * v
    a.x = fmaf(w.y, v.y, fmaf(w.x, v.x, a.x));
    a.y = fmaf(-w.y, v.x, fmaf(w.x, v.y, a.y));
}

// The same two as packed instructions.  A complex value is an aligned register pair (re, im); "w times v" is
//   (re, im) += w.re * (v.re, v.im)     and     (re, im) += w.im * (-v.im, v.re)   [conj(w): (+v.im, -v.re)],
// each ONE v_pk_fma_f32 when the half-swap and the sign are the instruction's own operand modifiers (op_sel / neg), which
// the compiler does not derive from C++ (it builds the swapped operand with moves): 2 instructions per tap instead of 4-6.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cmac_pk(v2f &a, v2f w, v2f v) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a) : "v"(w), "v"(v));
}
__device__ __forceinline__ void cmacc_pk(v2f &a, v2f w, v2f v) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
        : "+v"(a) : "v"(w), "v"(v));
}
__device__ __forceinline__ v2f as_v2f(float2 x) { return (v2f){x.x, x.y}; }

// sum over the Q adjacent lanes of a group (Q = 4, 8, 16; groups are aligned), in data-parallel-primitive moves
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

// magnitude re-projection (lwslib.cpp:356-360): target / |acc| as target * rsqrt(|acc|^2) with one Newton step (relative
// error < 2^-22); sums too small to square in fp32 are rescaled first so that "|acc| > 0" keeps the reference's meaning
__device__ __forceinline__ bool project(float2 acc, float target, float2 &v) {
    float m2 = acc.x * acc.x + acc.y * acc.y;
    const bool tiny = m2 < 1e-30f;
    const float ax = tiny ? acc.x * 0x1p60f : acc.x, ay = tiny ? acc.y * 0x1p60f : acc.y;
    m2 = tiny ? ax * ax + ay * ay : m2;
    float rs = __frsqrt_rn(m2);
    rs = rs * fmaf(-0.5f * m2 * rs, rs, 1.5f);
    const float sc = target * rs;
    v = make_float2(ax * sc, ay * sc);
    return m2 > 0.f;
}

// MAXT: launch bound (512 threads leave a lane 256 registers: the two windows, the weights and the sums fit without spills)
template <int Q, int L, bool SERIAL, int MAXT>
__global__ void __launch_bounds__(MAXT) k_online(OnlineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K1 = L + 1, WN = 2 * L + 2;
    constexpr int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2;                      // >= L + 2, even
    constexpr int DS_MIN = (SKB * (Q - 1) + L + 3) / 2;                         // 2 DS > SKB (Q-1) + L + 1
    static_assert(SKB >= L + 2 && 2 * DS_MIN > SKB * (Q - 1) + L + 1, "order-exact schedule");
    const int DS = a.DS;                                                        // >= DS_MIN (the launcher's choice)
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    const int NU = (F + 1) / 2;                                                 // pairs of bins per frame
    const int rps = LA + 1, per = a.n_thr + 1;
    const int nsweeps = T * per;
    float2 *S = reinterpret_cast<float2 *>(smem);                 // [NW][Np] (+ 2: the last window reads one column past a row)
    float *A = reinterpret_cast<float *>(S + (size_t)NW * Np + 2);  // [NW][Np]
    float2 *W = reinterpret_cast<float2 *>(A + (size_t)NW * Np + ((NW * Np) & 1));   // [3][Q][Q][K1]
    float2 *TW = W + 3 * Q * Q * K1;                                // [Q]
    float *thr_s = reinterpret_cast<float *>(TW + Q);               // [n_thr]
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    float2 *gS = a.state + (size_t)b * Tp * Np;
    const float *gA = a.amp + (size_t)b * Tp * Np;

    // weights; the self weight W[.][0][0] is not part of the sum (update type 2, lws.pyx:363)
    for (int i = tid; i < 3 * Q * Q * K1; i += nthr) {
        const int set = i / (Q * Q * K1), x = i % (Q * Q * K1);
        W[i] = (x % (Q * K1) == 0) ? make_float2(0.f, 0.f) : a.w[set][x];
    }
    if (tid < Q) TW[tid] = a.tw[tid];
    // ring slots that have not received a frame yet are read (with zero gain) by lanes whose right-hand frames do not
    // exist yet: they must hold finite numbers
    for (int i = tid; i < NW * Np + 2; i += nthr) S[i] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < a.n_thr; i += nthr) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    // rows 0 .. Q-1 (left edge pads and the first frame) are needed at step 0
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;
    for (int i = tid; i < loaded * Np; i += nthr) { S[i] = gS[i]; A[i] = gA[i]; }

    // this lane: side h (0: frame rho-r, 1: frame rho+r) of tap group r of frame position j of sweep slot sigma
    const int h = tid & 1, r = (tid >> 1) % Q, j = (tid / (2 * Q)) % rps, sigma = tid / (2 * Q * rps);
    const bool lane_used = sigma < NSW;
    int s = sigma;
    // per-sweep constants of the lane
    int rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0;
    int fb = 0, ctb = 0;                // LDS element offsets (column 0) of this lane's frame (rho-r or rho+r) and of frame rho
    bool valid = false, centre = false;
    float gain = 0.f, thr = 0.f;
    v2f w0[K1];                         // W[wset][0][r][k] (side 0) or its conjugate (side 1): see the sums below
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
        tstart = DS * s + SKS * rho;
        t_done = DS * s + SKS * m + NU - 1;     // last step of the sweep (its newest frame's last pair)
        const int e = rho + Q - 1;
        fb = ((h ? e + r : e - r) & (NW - 1)) * Np;
        ctb = (e & (NW - 1)) * Np;
        // Gains.  Lane (0,0): W[row][0][k] (S[c-k] + conj-weighted S[c+k]) of the centre frame if it takes part (the
        // asymmetric first estimate leaves it out, lwslib.cpp:1161-1178).  Lanes (r>=1, 0): the terms of frame rho-r, always;
        // lanes (r>=1, 1): those of frame rho+r if that frame is usable yet (r < ts; one-sided forms of lwslib.cpp:1222-1253
        // otherwise).
        if (h == 0) gain = (r == 0) ? (centre ? 1.f : 0.f) : 1.f;
        else gain = (r != 0 && r < ts) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k <= L; ++k) {
            const float2 w = W[(wset * Q + 0) * Q * K1 + r * K1 + k];
            w0[k] = (v2f){w.x, h ? -w.y : w.y};
        }
    };
    __syncthreads();
    setup();

    v2f wl[WN];                         // columns c-L .. c+L+1 of this lane's frame (lane (0,0): the centre frame)
#pragma unroll
    for (int i = 0; i < WN; ++i) wl[i] = (v2f){0.f, 0.f};

    const int t_end = DS * (nsweeps - 1) + SKS * (T - 1) + NU;
    int next_need = (loaded - (Q - 1)) * (DS * per + SKS);   // first step that touches row `loaded`: its frame's first sweep
    for (int t = 0; t < t_end; ++t) {
        const int u = t - tstart;
        if (valid && u >= 0 && u < NU) {
            const int c = 2 * u, n = c + L, e = rho + Q - 1;
            const bool has_b = c + 1 < F;
            const float2 zero = make_float2(0.f, 0.f);
            if constexpr (SERIAL) {
                if (r == 0 && h == 0) {
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int cb = c + bb, nb = n + bb;
                        if (cb >= F) break;
                        const int row = cb % Q, rowneg = (Q - row) % Q;
                        const float2 *wa = W + wset * Q * Q * K1 + row * Q * K1;
                        float2 acc = zero;
                        if (centre) {
                            const float2 *ctr = S + ctb + nb;
#pragma unroll
                            for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
                        }
#pragma unroll
                        for (int rr = 1; rr < Q; ++rr) {
                            const float2 *lf = S + ((e - rr) & (NW - 1)) * Np + nb;
                            const float2 *rt = S + ((e + rr) & (NW - 1)) * Np + nb;
                            const float2 *wa_r = W + wset * Q * Q * K1 + (row * Q + rr) * K1;
                            const float2 *wb_r = W + wset * Q * Q * K1 + (rowneg * Q + rr) * K1;
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
                        const int li = ctb + nb;
                        const float target = A[li];
                        if (target > thr) {
                            const float mag = sqrtf(acc.x * acc.x + acc.y * acc.y);
                            if (mag > 0.f) {
                                const float2 v = make_float2(acc.x * target / mag, acc.y * target / mag);
                                const float2 vc = make_float2(v.x, -v.y);
                                S[li] = v;
                                const int nyq = F + L - 1;     // Hermitian images in the pad columns (lwslib.cpp:362-367)
                                if (nb >= L + 1 && nb < 2 * L + 1) S[li + 2 * (L - nb)] = vc;
                                else if (nb >= F - 1 && nb < nyq) S[li + 2 * (nyq - nb)] = vc;
                            }
                        }
                    }
                }
            } else {
                // ---- the window: everything at the first pair of a frame, two new columns afterwards
                if (u == 0) {
#pragma unroll
                    for (int i = 0; i < WN; ++i) wl[i] = as_v2f(S[fb + i]);
                } else {
#pragma unroll
                    for (int i = 0; i < WN - 2; ++i) wl[i] = wl[i + 2];
                    wl[WN - 2] = as_v2f(S[fb + c + WN - 2]); wl[WN - 1] = as_v2f(S[fb + c + WN - 1]);
                }
                const float target_a = A[ctb + n], target_b = A[ctb + n + 1];   // (read with the window: off the dependent chain)
                // twiddles of the two bins: W[row][r][k] = W[0][r][k] tw^(row r); side 1 carries the conjugates
                float2 twa = TW[((c % Q) * r) & (Q - 1)], twb = TW[(((c + 1) % Q) * r) & (Q - 1)];
                if (h) { twa.y = -twa.y; twb.y = -twb.y; }
                // With w = W0[k] on side 0 and conj(W0[k]) on side 1 both sides form  sum w[k] X[c-k] + conj(w[k]) X[c+k]  over
                // their frame X -- side 0: W X(left, -k) + conj(W') X(left, +k), side 1: conj(W) X(right, -k) + W' X(right, +k),
                // the four kinds of term of lwslib.cpp:1182-1220 -- for the bins c (a) and c+1 (b); the term of bin c+1 that
                // reads column c waits for bin c's result
                v2f a14 = {0.f, 0.f}, b14 = {0.f, 0.f};
                cmac_pk(a14, w0[0], wl[L]);
                cmac_pk(b14, w0[0], wl[L + 1]);
#pragma unroll
                for (int k = 1; k <= L; ++k) {
                    cmac_pk(a14, w0[k], wl[L - k]);  cmacc_pk(a14, w0[k], wl[L + k]);
                    if (k >= 2) cmac_pk(b14, w0[k], wl[L + 1 - k]);
                    cmacc_pk(b14, w0[k], wl[L + 1 + k]);
                }
                // acc = sum over the 2Q lanes of  tw_lane * gain_lane * a_lane
                auto assemble = [&](v2f x, float2 tw) {
                    x *= gain;
                    float2 o;
                    o.x = fmaf(-tw.y, x.y, tw.x * x.x);
                    o.y = fmaf(tw.y, x.x, tw.x * x.y);
                    o.x = quad_sum<2 * Q>(o.x);
                    o.y = quad_sum<2 * Q>(o.y);
                    return o;
                };
                const float2 acc_a = assemble(a14, twa);
                if (r == 0 && h == 0) {
                    const int li = ctb + n;
                    const float target = target_a;
                    float2 v;
                    if (target > thr && project(acc_a, target, v)) {
                        const float2 vc = make_float2(v.x, -v.y);
                        S[li] = v;
                        wl[L] = as_v2f(v);
                        // Hermitian images: in the pad columns of the ring, and in this lane's own window while it covers them
                        // (bin c+1 of this very step reads such an image too -- column -c as its tap k = 2c+1, column
                        // 2(F-1)-c as its tap k = 2(F-1-c)-1 -- and its sums were formed above with the old value: add the change)
                        if (c >= 1 && c <= L) {
                            S[li - 2 * c] = vc;
#pragma unroll
                            for (int cc = 2; 2 * cc <= L; cc += 2)
                                if (c == cc) {
                                    if (2 * cc + 1 <= L) cmac_pk(b14, w0[2 * cc + 1], as_v2f(vc) - wl[L - 2 * cc]);
                                    wl[L - 2 * cc] = as_v2f(vc);
                                }
                        } else if (c >= F - 1 - L && c <= F - 2) {
                            S[li + 2 * (F - 1 - c)] = vc;
#pragma unroll
                            for (int d = 2; 2 * d + L <= WN - 1; d += 2)
                                if (c == F - 1 - d) {
                                    if (2 * d - 1 <= L) cmacc_pk(b14, w0[2 * d - 1], as_v2f(vc) - wl[2 * d + L]);
                                    wl[2 * d + L] = as_v2f(vc);
                                }
                        }
                    }
                }
                cmac_pk(b14, w0[1], wl[L]);       // (lane 0: the value just written; the others: unchanged column c of frame rho-r)
                const float2 acc_b = assemble(b14, twb);
                if (r == 0 && h == 0 && has_b) {
                    const int li = ctb + n + 1, cb = c + 1;
                    const float target = target_b;
                    float2 v;
                    if (target > thr && project(acc_b, target, v)) {
                        const float2 vc = make_float2(v.x, -v.y);
                        S[li] = v;
                        wl[L + 1] = as_v2f(v);
                        if (cb >= 1 && cb <= L) {
                            S[li - 2 * cb] = vc;
#pragma unroll
                            for (int cc = 1; 2 * cc - 1 <= L; cc += 2) if (cb == cc) wl[L + 1 - 2 * cc] = as_v2f(vc);
                        } else if (cb >= F - 1 - L && cb <= F - 2) {
                            S[li + 2 * (F - 1 - cb)] = vc;
#pragma unroll
                            for (int d = 1; 2 * d + L + 1 <= WN - 1; d += 2) if (cb == F - 1 - d) wl[2 * d + L + 1] = as_v2f(vc);
                        }
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
            next_need += DS * per + SKS;
        }
        __syncthreads();
    }
    // frames still in the ring
    const int first_row = loaded > NW ? loaded - NW : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e & (NW - 1)) * Np;
        for (int i = tid; i < Np; i += nthr) gS[(size_t)e * Np + i] = S[slot + i];
    }
}

// =====================================================================================================================
// Third layout (k_online3): one WAVE per tap group, one lane per (sweep slot, frame position).
//
// In the layout above every wave carries the whole dependent tail of a step -- cross-lane reduction, two re-projections,
// the image upkeep -- for one lane in 2Q: ~2/3 of the instructions a step issues.  Here wave w < 2Q-1 sums the taps of ONE
// frame offset (wave 0: the centre frame, waves 2r-1 / 2r: frames rho-r / rho+r) for all 64 (slot, position) units at once
// and leaves the two partial sums of a unit in LDS; the last wave adds them up, re-projects and writes.  The tap waves work
// ONE STEP AHEAD of it (double-buffered partial sums, one barrier per step), so the tail of step t overlaps the sums of step
// t+1.  That is order-exact because the values a tap wave reads one step early are final already -- every writer is at
// least L + 3 bins away from a window (2 DS >= SKB Q + 2 is required here) -- with three exceptions, which the tap waves
// leave out (zero window slots) and the last wave adds itself:
//   * frame rho-1 is SKB = L + 3 bins ahead: the last column of its window (bin c+1, tap +L) is being written;
//   * the centre frame's own recent outputs: columns c-2, c-1 (previous step), c (this step's first bin, read by the
//     second) and the Hermitian images of those three bins near the frame edges.  The centre wave reloads its window from
//     LDS every step (the unit's own writes land inside it), minus those columns.
// SERIAL (verification): the last wave sums every tap itself, from LDS, in the generic engine's order; same schedule.
// Waves of k_online3: 2Q-1 tap waves and the projection wave; Q = 4 adds an idle ninth wave so that the projection wave --
// the dependent chain every step waits for -- has a SIMD to itself (hardware waves w and w + 4 share one: it is wave 3, the
// idle one wave 7).
template <int Q> struct Online3Waves {
    static constexpr int N = (Q == 4) ? 9 : 2 * Q;
    static constexpr int PROJ = (Q == 8) ? 15 : 3, IDLE = (Q == 4) ? 7 : -1;
    static __host__ __device__ constexpr int tap_of(int hw) { return hw - (hw > PROJ ? 1 : 0) - (IDLE >= 0 && hw > IDLE ? 1 : 0); }
};

template <int Q, int L, bool SERIAL>
__global__ void __launch_bounds__(Online3Waves<Q>::N * 64) k_online3(OnlineArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K1 = L + 1, WN = 2 * L + 2, NTW = 2 * Q - 1;                  // NTW tap waves, then the projection wave
    constexpr int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2;
    static_assert(SKB >= L + 3, "the tap waves run one step ahead");
    const int DS = a.DS;
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, Np = F + 2 * L, Tp = T + 2 * (Q - 1), N = F - 1;
    const int NU = (F + 1) / 2;
    const int rps = LA + 1, per = a.n_thr + 1;
    const int nsweeps = T * per;
    // step types of the centre frame's edge terms: 0 none, 1..NLO the steps u = 1..NLO after a frame start, then N - c = 0..NHI-1
    constexpr int NLO = (L + 2) / 4, NHI = (L + 1) / 2 + 1, NST = 1 + NLO + NHI;
    float4 *P = reinterpret_cast<float4 *>(smem);                               // [2][NTW][64]: (sum of bin c, of bin c+1)
    float2 *ET = reinterpret_cast<float2 *>(P + 2 * NTW * 64);                  // [3][NST][6] (padded to 64 entries): edge-term weights
    float2 *S = ET + 192 + 64;                                                  // [NW][Np] (+ 2); the 64 entries below it: where image stores
                                                                                // of bins without an image go
    float *A = reinterpret_cast<float *>(S + (size_t)NW * Np + 2);              // [NW][Np]
    float2 *W = reinterpret_cast<float2 *>(A + (size_t)NW * Np + ((NW * Np) & 1));   // [3][Q][Q][K1]
    float2 *TW = W + 3 * Q * Q * K1;                                            // [Q]
    float *thr_s = reinterpret_cast<float *>(TW + Q);                           // [n_thr]
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wave = Online3Waves<Q>::tap_of(hw_wave);   // tap group of a tap wave
    float2 *gS = a.state + (size_t)b * Tp * Np;
    const float *gA = a.amp + (size_t)b * Tp * Np;

    for (int i = tid; i < 3 * Q * Q * K1; i += nthr) {
        const int x = i % (Q * Q * K1);
        W[i] = (x % (Q * K1) == 0) ? make_float2(0.f, 0.f) : a.w[i / (Q * Q * K1)][x];
    }
    if (tid < Q) TW[tid] = a.tw[tid];
    for (int i = tid; i < NW * Np + 2; i += nthr) S[i] = make_float2(0.f, 0.f);
    for (int i = tid; i < 2 * NTW * 64; i += nthr) P[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // Edge-term weights of the projection wave (see there): entry [wset][type][j], j = 0..2: what multiplies the conjugate of
    // the current value of bin c-j in the sum of bin c; 3, 4: of bins c-1, c-2 in the sum of bin c+1; 5: of (the new) bin c in
    // the sum of bin c+1.  Low edge (image column -y, tap k = c + y backwards): W[k]; high edge (column 2N - y, tap
    // k = 2(N-c) + d forwards): conj W[k].  W_ai (wset 1) has no centre term.
    if (tid < 3 * NST * 6) {
        const int ws = tid / (NST * 6), st = (tid / 6) % NST, jj = tid % 6;
        const float2 *wb = W + (ws * Q) * Q * K1;
        const int d = jj < 3 ? jj : (jj < 5 ? jj - 2 : 0), shift = jj < 3 ? 0 : 1;   // shift: the tap index moves by one for bin c+1
        float2 w = make_float2(0.f, 0.f);
        if (ws != 1 && st >= 1 && st <= NLO) {
            const int c = 2 * st, y = c - d, k = c + y + shift;
            if (y >= 1 && k <= L) w = wb[k];
        } else if (ws != 1 && st > NLO) {
            const int g = st - NLO - 1, k = 2 * g + d - shift;
            if (g + d >= 1 && g + d <= L && k >= 1 && k <= L) w = make_float2(wb[k].x, -wb[k].y);
        }
        ET[tid] = w;
    }
    if (tid < 64) S[-64 + tid] = make_float2(0.f, 0.f);
    __syncthreads();
    for (int i = tid; i < a.n_thr; i += nthr) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;
    for (int i = tid; i < loaded * Np; i += nthr) { S[i] = gS[i]; A[i] = gA[i]; }

    // this lane's unit: frame position j of sweep slot sigma; this wave's tap group: frame offset r, side h
    const int sigma = lane / rps, j = lane - sigma * rps;
    const bool lane_used = sigma < NSW;
    const bool is_proj = hw_wave == Online3Waves<Q>::PROJ, is_idle = hw_wave == Online3Waves<Q>::IDLE;
    const int r = (wave + 1) >> 1, h = (wave == 0) ? 0 : ((wave + 1) & 1);
    int s = sigma;
    int rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0;
    int fb = 0, ctb = 0, fbm1 = 0;
    bool valid = false, centre = false;
    float thr = 0.f;
    v2f w0[K1];                         // tap waves: W[wset][0][r][k] (side 0) or its conjugate (side 1)
    v2f twg[Q];                         // tap waves: gain * exp(2 pi j row r / Q) (conjugated on side 1), row = bin % Q
    v2f wc[K1];                         // projection wave: centre weights W[wset][0][0][k] (zero if the centre frame takes no part)
    v2f wlate = {0.f, 0.f};             // ... and conj W[wset][0][1][L]
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
        tstart = DS * s + SKS * rho;
        t_done = DS * s + SKS * m + NU - 1;
        const int e = rho + Q - 1;
        fb = ((h ? e + r : e - r) & (NW - 1)) * Np;
        ctb = (e & (NW - 1)) * Np;
        fbm1 = ((e - 1) & (NW - 1)) * Np;
        const float2 *wb = W + (wset * Q + 0) * Q * K1;
        if (is_proj) {
#pragma unroll
            for (int k = 0; k <= L; ++k) wc[k] = centre ? as_v2f(wb[k]) : (v2f){0.f, 0.f};
            const float2 wl_ = wb[1 * K1 + L];
            wlate = (v2f){wl_.x, -wl_.y};
        } else {
            float gain;
            if (h == 0) gain = (r == 0) ? (centre ? 1.f : 0.f) : 1.f;
            else gain = (r != 0 && r < ts) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k <= L; ++k) {
                const float2 w = wb[r * K1 + k];
                w0[k] = (v2f){w.x, h ? -w.y : w.y};
            }
#pragma unroll
            for (int row = 0; row < Q; ++row) {
                const float2 tw = TW[(row * r) & (Q - 1)];
                twg[row] = (v2f){gain * tw.x, gain * (h ? -tw.y : tw.y)};
            }
        }
    };
    __syncthreads();
    setup();

    const int t_end = DS * (nsweeps - 1) + SKS * (T - 1) + NU;
    int next_need = (loaded - (Q - 1)) * (DS * per + SKS);
    // bring in the next frame before the iteration in which a tap wave first touches it (the step after next)
    auto load_frames = [&](int t) {
        while (loaded < T + Q - 1 && next_need <= t + 2) {
            const int slot = (loaded & (NW - 1)) * Np;
            const bool evict = loaded >= NW;
            for (int i = tid; i < Np; i += nthr) {
                if (evict) gS[(size_t)(loaded - NW) * Np + i] = S[slot + i];
                S[slot + i] = gS[(size_t)loaded * Np + i];
                A[slot + i] = gA[(size_t)loaded * Np + i];
            }
            ++loaded;
            next_need += DS * per + SKS;
        }
    };

    // the sums of one step of a tap wave; KIND 0: frames rho-+r; 1: frame rho-1 (the last column of its window is still being
    // written: left out); 2: the centre frame (its unit's own recent outputs are left out)
    auto tap_loop = [&](auto kind_c) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        v2f wl[WN];
#pragma unroll
        for (int i = 0; i < WN; ++i) wl[i] = (v2f){0.f, 0.f};
        LAB(unsigned long long lab_acc[4] = {0, 0, 0, 0};)
        for (int t = -1; t < t_end; ++t) {
            const int tt = t + 1;               // the step these waves prepare
            const int u = tt - tstart;
            LAB(const unsigned long long lt0 = lab_now(); unsigned long long lt1 = lt0;)
            if (!SERIAL && valid && u >= 0 && u < NU) {
                const int c = 2 * u;
                // the window, columns c-L .. c+L+1 of this wave's frame, fresh from LDS (whatever is written concurrently is
                // among the columns left out)
                const float2 *src = S + fb + c;
                if (KIND == 2 || u == 0) {
#pragma unroll
                    for (int i = 0; i < WN; ++i) {
                        const bool skip = (KIND == 1 && i == WN - 1) || (KIND == 2 && i >= L - 2 && i <= L);
                        wl[i] = skip ? (v2f){0.f, 0.f} : as_v2f(src[i]);
                    }
                } else {   // the window slides by two columns per step (no writer comes near it: see the header)
                    constexpr int LG = KIND == 1 ? 1 : 0;
#pragma unroll
                    for (int i = 0; i < WN - 2; ++i) wl[i] = wl[i + 2];
                    wl[WN - 2 - LG] = as_v2f(src[WN - 2 - LG]);
                    wl[WN - 1 - LG] = as_v2f(src[WN - 1 - LG]);
                    if (LG) wl[WN - 1] = (v2f){0.f, 0.f};
                }
                if constexpr (KIND == 2) {
                    // images of the bins y = c-d (d = 0, 1, 2) inside the window: column -y (1 <= y) at slot L - 2c + d, column
                    // 2N - y (N-L <= y <= N-1) at slot L + 2(N-c) + d -- a handful of (step, slot) pairs, spelled out
                    const int g = N - c;
                    if (2 * c <= L + 2 || 2 * g <= L + 1) {
                        static_for<(L + 2) / 4>([&](auto iu) {
                            constexpr int U = decltype(iu)::value + 1, C = 2 * U;
                            if (u == U) {
                                static_for<3>([&](auto id) {
                                    constexpr int D = decltype(id)::value, I = L - 2 * C + D;
                                    if constexpr (C - D >= 1 && I >= 0 && I < WN) wl[I] = (v2f){0.f, 0.f};
                                });
                            }
                        });
                        static_for<(L + 1) / 2 + 1>([&](auto ig) {
                            constexpr int G = decltype(ig)::value;
                            if (g == G) {
                                static_for<3>([&](auto id) {
                                    constexpr int D = decltype(id)::value, I = L + 2 * G + D;
                                    if constexpr (G + D >= 1 && G + D <= L && I < WN) wl[I] = (v2f){0.f, 0.f};
                                });
                            }
                        });
                    }
                }
                LAB(lt1 = lab_now();)
                // sum w[k] X[c-k] + conj(w[k]) X[c+k] over this wave's frame X, for the bins c (a) and c+1 (b)
                v2f a14 = {0.f, 0.f}, b14 = {0.f, 0.f};
                if constexpr (KIND != 2) {       // (the centre frame's own bin is not a tap)
                    cmac_pk(a14, w0[0], wl[L]);
                    cmac_pk(b14, w0[0], wl[L + 1]);
                }
#pragma unroll
                for (int k = 1; k <= L; ++k) {
                    cmac_pk(a14, w0[k], wl[L - k]);  cmacc_pk(a14, w0[k], wl[L + k]);
                    cmac_pk(b14, w0[k], wl[L + 1 - k]);  cmacc_pk(b14, w0[k], wl[L + 1 + k]);
                }
                v2f twa = twg[0], twb = twg[1 & (Q - 1)];
#pragma unroll
                for (int row = 2; row < Q; row += 2) {
                    if ((c & (Q - 1)) == row) { twa = twg[row]; twb = twg[row + 1]; }
                }
                v2f pa = {0.f, 0.f}, pb = {0.f, 0.f};
                cmac_pk(pa, twa, a14);
                cmac_pk(pb, twb, b14);
                P[((tt & 1) * NTW + wave) * 64 + lane] = make_float4(pa.x, pa.y, pb.x, pb.y);
            }
            LAB(const unsigned long long lt2 = lab_now();)
            if (tt >= t_done) { s += NSW; setup(); }
            load_frames(t);
            LAB(const unsigned long long lt3 = lab_now();)
            __syncthreads();
            LAB(const unsigned long long lt4 = lab_now(); lab_acc[0] += lt1 - lt0; lab_acc[1] += lt2 - lt1; lab_acc[2] += lt3 - lt2; lab_acc[3] += lt4 - lt3;)
        }
        LAB(if (b == 0 && lane == 0) { for (int i = 0; i < 4; ++i) g_lab[8 + hw_wave * 8 + i] = lab_acc[i]; })
    };
    if (is_idle) {
        for (int t = -1; t < t_end; ++t) { load_frames(t); __syncthreads(); }
    } else if (!is_proj) {
        if (wave == 0) tap_loop(std::integral_constant<int, 2>{});
        else if (wave == 1) tap_loop(std::integral_constant<int, 1>{});
        else tap_loop(std::integral_constant<int, 0>{});
    } else {
        // ------------------------------------------------------------------------------------------ projection wave
        asm volatile("s_setprio 3");            // the dependent chain of a step: ahead of the tap waves of its SIMD
        v2f p1 = {0.f, 0.f}, p2 = {0.f, 0.f};   // current values of columns c-1, c-2 of the unit's frame
        LAB(unsigned long long lab_acc[4] = {0, 0, 0, 0};)
        for (int t = -1; t < t_end; ++t) {
            const int u = t - tstart;
            LAB(const unsigned long long lt0 = lab_now(); unsigned long long lt1 = lt0;)
            if (t >= 0 && valid && u >= 0 && u < NU) {
                const int c = 2 * u, n = c + L;
                const bool has_b = c + 1 < F;
                const int li = ctb + n;
                if constexpr (SERIAL) {
                    const int e = rho + Q - 1;
                    const float2 zero = make_float2(0.f, 0.f);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int cb = c + bb, nb = n + bb;
                        if (cb >= F) break;
                        const int row = cb % Q, rowneg = (Q - row) % Q;
                        const float2 *wa = W + wset * Q * Q * K1 + row * Q * K1;
                        float2 acc = zero;
                        if (centre) {
                            const float2 *ctr = S + ctb + nb;
#pragma unroll
                            for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
                        }
#pragma unroll
                        for (int rr = 1; rr < Q; ++rr) {
                            const float2 *lf = S + ((e - rr) & (NW - 1)) * Np + nb;
                            const float2 *rt = S + ((e + rr) & (NW - 1)) * Np + nb;
                            const float2 *wa_r = W + wset * Q * Q * K1 + (row * Q + rr) * K1;
                            const float2 *wb_r = W + wset * Q * Q * K1 + (rowneg * Q + rr) * K1;
                            const bool two = rr < ts;
                            pair(acc, wa_r[0], lf[0], two ? rt[0] : zero);
#pragma unroll
                            for (int k = 1; k <= L; ++k) {
                                pair(acc, wa_r[k], lf[-k], two ? rt[-k] : zero);
                                pair(acc, wb_r[k], two ? rt[k] : zero, lf[k]);
                            }
                        }
                        const int lj = ctb + nb;
                        const float target = A[lj];
                        if (target > thr) {
                            const float mag = sqrtf(acc.x * acc.x + acc.y * acc.y);
                            if (mag > 0.f) {
                                const float2 v = make_float2(acc.x * target / mag, acc.y * target / mag);
                                const float2 vc = make_float2(v.x, -v.y);
                                S[lj] = v;
                                const int nyq = F + L - 1;
                                if (nb >= L + 1 && nb < 2 * L + 1) S[lj + 2 * (L - nb)] = vc;
                                else if (nb >= F - 1 && nb < nyq) S[lj + 2 * (nyq - nb)] = vc;
                            }
                        }
                    }
                } else {
                    // everything the step reads, up front: one wait
                    const float4 *pp = P + (t & 1) * NTW * 64 + lane;
                    float4 part[NTW];
#pragma unroll
                    for (int w = 0; w < NTW; ++w) part[w] = pp[w * 64];
                    const v2f oldA = as_v2f(S[li]), oldB = as_v2f(S[li + 1]);
                    const float target_a = A[li], target_b = A[li + 1];
                    const v2f xlate = as_v2f(S[fbm1 + c + 1 + 2 * L]);
                    const v2f twl = as_v2f(TW[(c + 1) & (Q - 1)]);
                    const v2f im1 = as_v2f(S[ctb + L - 1]), im2 = as_v2f(S[ctb + L - 2]);   // columns -1, -2 (images): what a frame starts with
                    const int g = N - c;
                    const int stype = (u >= 1 && u <= NLO) ? u : (g < NHI ? NLO + 1 + g : 0);
                    const float4 *et = reinterpret_cast<const float4 *>(ET + (wset * NST + stype) * 6);
                    const float4 e01 = et[0], e23 = et[1], e45 = et[2];
                    LAB(lt1 = lab_now();)
                    if (u == 0) { p1 = im1; p2 = im2; }
                    v2f accA = {part[0].x, part[0].y}, accB = {part[0].z, part[0].w};
#pragma unroll
                    for (int w = 1; w < NTW; ++w) {
                        accA += (v2f){part[w].x, part[w].y};
                        accB += (v2f){part[w].z, part[w].w};
                    }
                    // frame rho-1, bin c+1, tap +L: the column that was being written while the tap wave summed
                    {
                        v2f x = {0.f, 0.f};
                        cmac_pk(x, wlate, xlate);
                        cmac_pk(accB, twl, x);
                    }
                    // the centre frame's columns c-1, c-2 (and c, below): the unit's own last outputs
                    cmac_pk(accA, wc[1], p1);
                    cmac_pk(accA, wc[2], p2);
                    cmac_pk(accB, wc[2], p1);
                    if (L >= 3) cmac_pk(accB, wc[L >= 3 ? 3 : 0], p2);
                    // ... and their Hermitian images near the frame edges: the image of bin y = c-d is column -y, tap k = c + y of
                    // bin c (k + 1 of bin c+1), or column 2N - y, tap k = 2(N-c) + d forwards (k - 1 of bin c+1) -- a handful
                    // of (step, tap) pairs.  Their weights come from a table indexed by the kind of step (zeros for all but ~4
                    // steps of a frame), so that the chain every step waits for has no branch here.
                    {
                        const v2f cjA = {oldA.x, -oldA.y}, cj1 = {p1.x, -p1.y}, cj2 = {p2.x, -p2.y};
                        cmac_pk(accA, (v2f){e01.x, e01.y}, cjA);
                        cmac_pk(accA, (v2f){e01.z, e01.w}, cj1);
                        cmac_pk(accA, (v2f){e23.x, e23.y}, cj2);
                        cmac_pk(accB, (v2f){e23.z, e23.w}, cj1);
                        cmac_pk(accB, (v2f){e45.x, e45.y}, cj2);
                    }
                    // ---- first bin
                    v2f newA;
                    {
                        float m2 = accA.x * accA.x + accA.y * accA.y;
                        v2f q = accA;
                        if (m2 < 1e-30f) {           // too small to square in fp32 (or zero): rescale, so that "|acc| > 0" keeps its meaning
                            q *= 0x1p60f;
                            m2 = q.x * q.x + q.y * q.y;
                        }
                        const float sc = target_a * __frsqrt_rn(m2);
                        const bool upd = target_a > thr && m2 > 0.f;
                        newA = upd ? q * sc : oldA;
                    }
                    cmac_pk(accB, wc[1], newA);
                    cmac_pk(accB, (v2f){e45.z, e45.w}, (v2f){newA.x, -newA.y});   // the image of bin c itself, as the second bin sees it
                    // ---- second bin
                    v2f newB;
                    {
                        float m2 = accB.x * accB.x + accB.y * accB.y;
                        v2f q = accB;
                        if (m2 < 1e-30f) {
                            q *= 0x1p60f;
                            m2 = q.x * q.x + q.y * q.y;
                        }
                        const float sc = target_b * __frsqrt_rn(m2);
                        const bool upd = has_b && target_b > thr && m2 > 0.f;
                        newB = upd ? q * sc : oldB;
                    }
                    // unchanged bins are written back as they were; Hermitian images in the pad columns (lwslib.cpp:362-367)
                    // (a bin without an image stores into a spare slot: no branch)
                    const int spare = lane - 64, cb = c + 1;
                    const int ia = (c >= 1 && c <= L) ? li - 2 * c : ((c >= N - L && c <= N - 1) ? li + 2 * (N - c) : spare);
                    const int ib = (cb <= L) ? li + 1 - 2 * cb : ((cb >= N - L && cb <= N - 1) ? li + 1 + 2 * (N - cb) : spare);
                    S[li] = make_float2(newA.x, newA.y);
                    S[has_b ? li + 1 : spare] = make_float2(newB.x, newB.y);
                    S[ia] = make_float2(newA.x, -newA.y);
                    S[ib] = make_float2(newB.x, -newB.y);
                    p2 = newA;
                    p1 = newB;
                }
            }
            LAB(const unsigned long long lt2 = lab_now();)
            if (t >= t_done) { s += NSW; setup(); }
            load_frames(t);
            LAB(const unsigned long long lt3 = lab_now();)
            __syncthreads();
            LAB(const unsigned long long lt4 = lab_now(); lab_acc[0] += lt1 - lt0; lab_acc[1] += lt2 - lt1; lab_acc[2] += lt3 - lt2; lab_acc[3] += lt4 - lt3;)
        }
        LAB(if (b == 0 && lane == 0) { for (int i = 0; i < 4; ++i) g_lab[8 + hw_wave * 8 + i] = lab_acc[i]; g_lab[0] = t_end + 1; })
    }
    const int first_row = loaded > NW ? loaded - NW : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e & (NW - 1)) * Np;
        for (int i = tid; i < Np; i += nthr) gS[(size_t)e * Np + i] = S[slot + i];
    }
}


// =====================================================================================================================
// Fourth layout (k_online4; default when it fits).  k_online3 is bound by vector-ALU issue on the SIMD that carries three of
// its seven tap waves (profiles/r03_pmc_sq_online*.json: 863 vector instructions per step and workgroup, of which 7 x 32 are
// the register moves of the sliding windows and ~7 x 26 index / predicate arithmetic).  Same roles -- one wave per tap group,
// one projection wave, tap waves one step ahead -- with the per-step overhead removed:
//   * The step loop is unrolled by two and a neighbour-frame tap wave works on a TWO-step window of 2L + 4 columns whose
//     register names are fixed: no window slides.  Rows of the LDS ring have an even stride, windows start at even columns:
//     every window read is an aligned 16-byte cell.  At the end of a pair the next pair's first cells -- values the wave
//     already holds -- are simply read again under their new names (LDS has the bandwidth to spare, the vector ALU has not);
//     that read is issued before the barrier.
//   * With an EVEN lag between sweeps (DS even; SKS is even already) every lane of the workgroup is at an even bin pair
//     u = t - tstart in even steps and at an odd one in odd steps: the twiddle of a bin (bin mod Q) is static and a lane that
//     starts a frame needs no special case.  The smallest order-exact lag is odd, though (SKS Q + 1), and the lag is the
//     length of the chain: Q = 2, Q = 4 and the table-twiddle variant take an ODD lag when that is the smallest.  Every other
//     sweep then starts at an odd step; its lanes pick their twiddles two rows further on and re-read the few window cells
//     that hold Hermitian images stored too late for the early read (tap_loop; tools/online_schedule_check.py is the model
//     of who stores what when, tests/test_online_schedule.py runs it, the GPU tests compare the two lags bit for bit).
//     LWS_ONLINE_LAG_PLUS=k adds k steps to the lag (comparison runs).
//   * tap waves carry no validity predicate: a lane without work computes on clamped addresses and nobody reads its sums.
//   * 2Q waves, two per SIMD for Q = 4: the centre-frame wave shares the projection wave's SIMD (no idle wave).
//   * the ring holds NWR frames (run-time, not a power of two), as many as the look-ahead needs: 2048-point frames fit.
// SERIAL (verification): as in k_online3, the projection wave sums every tap itself in the generic engine's order.
template <int Q> struct Online4Waves {
#ifndef LWS_ONLINE4_IDLE_WAVE   // (tried: an idle wave on the projection wave's SIMD and the centre wave elsewhere: 57.8 vs 50.9 ms)
    static constexpr int N = 2 * Q;
    static constexpr int PROJ = 3, CENTRE = (Q == 2) ? 2 : (Q == 3 ? 5 : 7), IDLE = -1;   // (hardware waves; N >= 8 from Q = 4 on)
#else
    static constexpr int N = (Q == 4) ? 9 : 2 * Q;
    static constexpr int PROJ = 3, CENTRE = (Q == 2) ? 2 : (Q == 4 ? 8 : 7), IDLE = (Q == 4) ? 7 : -1;
#endif
    static __host__ __device__ constexpr int tap_of(int hw) {
        return hw == CENTRE ? 0 : hw - (hw > PROJ ? 1 : 0) - (hw > CENTRE ? 1 : 0) - (IDLE >= 0 && hw > IDLE ? 1 : 0) + 1;
    }
};

// w * v and conj(w) * v as one v_pk_mul_f32 + one v_pk_fma_f32 (no zeroed accumulator)
__device__ __forceinline__ v2f cmul_pk(v2f w, v2f v) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r) : "v"(w), "v"(v));
    return r;
}
__device__ __forceinline__ v2f cmulc_pk(v2f w, v2f v) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(w), "v"(v));
    return r;
}
// a += w * conj(v)
__device__ __forceinline__ void cmac_cv_pk(v2f &a, v2f w, v2f v) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "+v"(a) : "v"(w), "v"(v));
}

// acc += M (px, py)^T with the 2 x 2 real matrix M = (col0 | col1) held as (col0.x, col0.y, col1.x, col1.y): what
// w p + e conj(p) is for complex w, e -- col0 = (w.x + e.x, w.y + e.y), col1 = (e.y - w.y, w.x - e.x) -- in two instructions
__device__ __forceinline__ void mat_mac_pk(v2f &a, float4 M, v2f p) {
    const v2f c0 = {M.x, M.y}, c1 = {M.z, M.w};
    asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(a) : "v"(c0), "v"(c1), "v"(p));
}
__device__ __forceinline__ v2f mat_mul_pk(float4 M, v2f p) {
    const v2f c0 = {M.x, M.y}, c1 = {M.z, M.w};
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "=&v"(r) : "v"(c0), "v"(c1), "v"(p));
    return r;
}
// the power of two that brings a positive normal number to [1, 2) (1 for zero, denormals, infinities)
__device__ __forceinline__ float pow2_to_unit(float amax) {
    const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;
    return (e == 0u || e == 0xffu) ? 1.0f : __uint_as_float((e >= 254u ? 1u : 254u - e) << 23);
}

// BIG: frames too long for the ring to hold their target magnitudes and the step table as well (4096-point STFTs: a ring of eight
// 2060-column frames is 132 KB of values).  The magnitudes are then read from the caller's buffer, one step ahead (a global load on
// the projection wave's chain: slower steps, 10x faster than the generic engine such frames used to get), and the step table's
// entries are computed instead of fetched.  Production variant only.
// TWT: the twiddle of a bin comes from a table in LDS (per lane: row = bin mod PT) instead of being static in the two-step loop
// ODD: the build for an odd lag between sweeps (the even-lag build carries none of its tests)
template <int Q, int L, bool SERIAL, bool BIG = false, bool TWT = false, bool ODD = false>
__global__ void __launch_bounds__(Online4Waves<Q>::N * 64) k_online4(OnlineArgs a) {
    static_assert(!ODD || (!SERIAL && (TWT || Q == 2 || Q == 4)), "odd lags: the production variants of Q = 2, 4 and the table-twiddle one");
    static_assert(!(BIG && SERIAL), "the verification variant keeps everything in LDS");
    static_assert(!(TWT && (SERIAL || BIG)) && (TWT || Q == 2 || Q == 4 || Q == 8) && Q >= 2 && Q <= 8, "table twiddles: production variant");
    constexpr int TQ = Q <= 4 ? 4 : 8;                  // twiddles per row of the table (frame offsets 0 .. TQ - 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int K1 = L + 1, WN = 2 * L + 2, NTW = 2 * Q - 1;
    constexpr int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2;
    constexpr int NCELL = WN / 2 + 1;                   // 16-byte cells (two columns) of a two-step window
    static_assert(SKB >= L + 3 && (SKS & 1) == 0 && (WN & 3) == 0, "two-step windows of aligned cells");
    const int DS = a.DS;                                // (even, or odd: see above)
    const int F = a.F, T = a.T, LA = a.LA, NSW = a.NSW, Tp = T + 2 * (Q - 1), N = F - 1;
    const int NWR = a.NWR, NPS = a.NPS;                 // (a row of the ring: F + 2 L columns, rounded up to even)
    const int NU = (F + 1) / 2;
    const int rps = LA + 1, per = a.n_thr + 1;
    const int nsweeps = T * per;
    constexpr int NLO = (L + 2) / 4, NHI = (L + 1) / 2 + 1, NST = 1 + NLO + NHI;
    // LDS layout (byte offsets; lds_of in shape4_of is the same sum)
    const unsigned oET = 2u * NTW * 64 * 16, oS = oET + (224 + 64) * 8, oA = oS + ((unsigned)NWR * NPS + 8) * 8, oW = oA + (BIG ? 0u : (unsigned)NWR * NPS * 4),
                   oTW = oW + 3u * Q * Q * K1 * 8, oThr = oTW + Q * 8, oTAB = (oThr + (unsigned)a.n_thr * 4 + 15u) & ~15u,
                   oTT = oTAB + (BIG ? 0u : (unsigned)((F + 1) / 2) * 16u);
    float4 *P = reinterpret_cast<float4 *>(smem);                               // [2][NTW][64]: (sum of bin c, of bin c+1)
    float4 *MT = reinterpret_cast<float4 *>(smem + oET);                        // [3][NST][6]: own-history matrices of the projection wave (below)
    static_assert(3 * NST * 6 * 16 <= 224 * 8, "matrix table");
    float2 *S = reinterpret_cast<float2 *>(smem + oS);                          // [NWR][NPS] (+ 8); the 64 entries below it: spare slots
    float *A = reinterpret_cast<float *>(smem + oA);                            // [NWR][NPS]
    float2 *W = reinterpret_cast<float2 *>(smem + oW);                          // [3][Q][Q][K1]
    float2 *TW = reinterpret_cast<float2 *>(smem + oTW);                        // [Q]
    float *thr_s = reinterpret_cast<float *>(smem + oThr);                      // [n_thr]
    int4 *TAB = reinterpret_cast<int4 *>(smem + oTAB);                          // [NU] step table of the projection wave, by bin pair
    const float2 *TT = reinterpret_cast<const float2 *>(smem + oTT);            // TWT: [PT + 3][TQ] twiddles by bin mod PT and frame offset
    const int PT = TWT ? a.PT : 1;
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool is_proj = hw_wave == Online4Waves<Q>::PROJ;
    const int wave = Online4Waves<Q>::tap_of(hw_wave);   // tap group of a tap wave (0: the centre frame)
    // the caller's extended buffers: Lu <= L pad columns on either side (bin c of the kernel's rows is column c + L, of the
    // caller's c + Lu: its columns sit dL further right in a row of the ring; the outer 2 dL columns only ever meet zero weights)
    const int Npu = F + 2 * a.Lu, dL = L - a.Lu, K1u = a.Lu + 1;
    float2 *gS = a.state + (size_t)b * Tp * Npu;
    const float *gA = a.amp + (size_t)b * Tp * Npu;

    // The weighted sums are only ever normalised, so the weights of this spectrogram are scaled by the power of two that
    // brings its largest target magnitude to [1, 2) -- exact -- and |sum|^2 can be formed in fp32 for data of any scale
    // without the rescue path (rescale and square again when the square underflows) both re-projections of a step carried.
    // (A sum below 1e-19 of the data's scale now counts as zero -- no update -- where the reference, in fp64, would still
    // take its phase; fp32 rounding decides that phase long before.)  Not in the verification variant: bit for bit the
    // generic engine.
    float wscale = 1.0f;
    if constexpr (!SERIAL) {
        float mx = 0.f;
        for (int i = tid; i < Tp * Npu; i += nthr) mx = fmaxf(mx, gA[i]);
        float *red = reinterpret_cast<float *>(smem);                // (P: not in use yet)
        red[tid] = mx;
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < nthr; ++i) mx = fmaxf(mx, red[i]);
            red[0] = pow2_to_unit(mx);
        }
        __syncthreads();
        wscale = red[0];
        __syncthreads();
    }
    for (int i = tid; i < 3 * Q * Q * K1; i += nthr) {
        const int x = i % (Q * Q * K1), pr = x / K1, k = x - pr * K1;
        const float2 w = k < K1u ? a.w[i / (Q * Q * K1)][pr * K1u + k] : make_float2(0.f, 0.f);
        W[i] = (x % (Q * K1) == 0) ? make_float2(0.f, 0.f) : make_float2(w.x * wscale, w.y * wscale);
    }
    if (tid < Q) TW[tid] = a.tw[tid];
    if constexpr (TWT) for (int i = tid; i < (PT + 3) * TQ; i += nthr) reinterpret_cast<float2 *>(smem + oTT)[i] = a.twt[i];
    for (int i = tid; i < NWR * NPS + 8; i += nthr) S[i] = make_float2(0.f, 0.f);
    if constexpr (!BIG) for (int i = tid; i < NWR * NPS; i += nthr) A[i] = 0.f;
    for (int i = tid; i < 2 * NTW * 64; i += nthr) P[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // Own-history terms of the projection wave, per weight set and kind of step (0: none of the edge terms; 1..NLO: the steps
    // u = 1..NLO after a frame start; then N - c = 0..NHI-1), as 2 x 2 matrices (mat_mac_pk): entry 0 / 1: what bins c-1 / c-2
    // contribute to the sum of bin c -- the centre weight W[k] times the value plus the edge weight times its conjugate
    // (the Hermitian image of that bin as a tap near a frame edge: low edge, image column -y, tap k = c + y backwards: W[k];
    // high edge, column 2N - y, tap k = 2(N-c) + d forwards: conj W[k]); 2 / 3: the same two bins in the sum of bin c+1;
    // 4: the new bin c in the sum of bin c+1; 5: (x, y) = weight of the conjugate of the OLD value of bin c in the sum of
    // bin c (its own image).  W_ai (set 1: first estimate of a frame) has no centre term.
    if (tid < 3 * NST) {
        const int ws = tid / NST, st = tid % NST;
        const float2 *wb = W + (ws * Q) * Q * K1;
        auto edge = [&](int jj) {   // jj = 0..2: conj of bin c-jj in the sum of bin c; 3, 4: of bins c-1, c-2 in the sum of bin c+1; 5: of the new bin c there
            const int d = jj < 3 ? jj : (jj < 5 ? jj - 2 : 0), shift = jj < 3 ? 0 : 1;
            float2 w = make_float2(0.f, 0.f);
            if (ws != 1 && st >= 1 && st <= NLO) {
                const int c = 2 * st, y = c - d, k = c + y + shift;
                if (y >= 1 && k <= L) w = wb[k];
            } else if (ws != 1 && st > NLO) {
                const int g = st - NLO - 1, k = 2 * g + d - shift;
                if (g + d >= 1 && g + d <= L && k >= 1 && k <= L) w = make_float2(wb[k].x, -wb[k].y);
            }
            return w;
        };
        auto centre_w = [&](int k) { return (ws != 1 && k <= L) ? wb[k] : make_float2(0.f, 0.f); };
        auto mat = [](float2 w, float2 e) { return make_float4(w.x + e.x, w.y + e.y, e.y - w.y, w.x - e.x); };
        float4 *mt = MT + (size_t)tid * 6;
        mt[0] = mat(centre_w(1), edge(1));
        mt[1] = mat(centre_w(2), edge(2));
        mt[2] = mat(centre_w(2), edge(3));
        mt[3] = mat(centre_w(3), edge(4));
        mt[4] = mat(centre_w(1), edge(5));
        const float2 e0 = edge(0);
        mt[5] = make_float4(e0.x, e0.y, 0.f, 0.f);
    }
    if (tid < 64) S[-64 + tid] = make_float2(0.f, 0.f);
    // step table: x = byte offset of the pair's matrices in MT (within a weight set), y / z = byte offsets of the Hermitian
    // images of the pair's bins relative to the bins themselves (0: no image), w = sign bit if the pair has no second bin
    for (int uu = tid; uu < NU; uu += nthr) {
        const int c = 2 * uu, cb = c + 1, g = N - c;
        const int stype = (uu >= 1 && uu <= NLO) ? uu : (g < NHI ? NLO + 1 + g : 0);
        const int da = (c >= 1 && c <= L) ? -16 * c : ((c >= N - L && c <= N - 1) ? 16 * (N - c) : 0);
        const int db = (cb <= L) ? -16 * cb : ((cb >= N - L && cb <= N - 1) ? 16 * (N - cb) : 0);
        if constexpr (!BIG) TAB[uu] = make_int4(stype * 96, da, db, cb < F ? 0 : (int)0x80000000);
    }
    __syncthreads();
    for (int i = tid; i < a.n_thr; i += nthr) thr_s[i] = a.thr[(size_t)b * a.n_thr + i];
    int loaded = Q < T + Q - 1 ? Q : T + Q - 1;
    for (int r0 = 0; r0 < loaded; ++r0)
        for (int i = tid; i < Npu; i += nthr) {
            S[r0 * NPS + dL + i] = gS[(size_t)r0 * Npu + i];
            if constexpr (!BIG) A[r0 * NPS + dL + i] = gA[(size_t)r0 * Npu + i];
        }

    const int sigma = lane / rps, j = lane - sigma * rps;
    const bool lane_used = sigma < NSW;
    const int r = (wave + 1) >> 1, h = (wave == 0) ? 0 : ((wave + 1) & 1);
    int s = sigma;
    int rho = 0, tstart = 0, t_done = 0, ts = 1, wset = 0;
    int fb = 0, ctb = 0, fbm1 = 0;
    int e_own = 0;                      // BIG: the lane's frame (row of the caller's magnitude buffer)
    bool valid = false, centre = false;
    float thr = 0.f;
    v2f w0[K1];                         // tap waves: W[wset][0][r][k] (side 0) or its conjugate (side 1)
    v2f twg[Q];                         // tap waves: gain * exp(2 pi j row r / Q) (conjugated on side 1), row = bin % Q
    v2f gvec = {0.f, 0.f};              // TWT, tap waves: (gain, +-gain): what turns a table twiddle into this wave's (conjugated on side 1)
    v2f wc[K1];                         // projection wave: centre weights W[wset][0][0][k] (zero if the centre frame takes no part)
    v2f wlate = {0.f, 0.f};             // ... and conj W[wset][0][1][L]
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
        tstart = DS * s + SKS * rho;
        t_done = DS * s + SKS * m + NU - 1;     // (even when DS is: SKS is, NU is odd)
        const int e = rho + Q - 1;
        e_own = e;
        fb = ((h ? e + r : e - r) % NWR) * NPS;
        ctb = (e % NWR) * NPS;
        fbm1 = ((e - 1) % NWR) * NPS;
        const float2 *wb = W + (wset * Q + 0) * Q * K1;
        if (is_proj) {
#pragma unroll
            for (int k = 0; k <= L; ++k) wc[k] = centre ? as_v2f(wb[k]) : (v2f){0.f, 0.f};
            const float2 wl_ = wb[1 * K1 + L];
            wlate = (v2f){wl_.x, -wl_.y};
        } else {
            float gain;
            if (h == 0) gain = (r == 0) ? (centre ? 1.f : 0.f) : 1.f;
            else gain = (r != 0 && r < ts) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k <= L; ++k) {
                const float2 w = wb[r * K1 + k];
                w0[k] = (v2f){w.x, h ? -w.y : w.y};
                if (r == 0) w0[k] *= (v2f){gain, gain};       // the centre-frame wave multiplies by no twiddle: its gain (0 or 1) goes here
            }
            gvec = (v2f){gain, h ? -gain : gain};
            if constexpr (!TWT) {
                // (ODD: a lane that starts at an odd step is at an odd bin pair in even steps: its bins are two rows further on)
                const int shift = ODD ? 2 * (tstart & 1) : 0;
#pragma unroll
                for (int row = 0; row < Q; ++row) {
                    const float2 tw = TW[((row + shift) * r) & (Q - 1)];
                    twg[row] = (v2f){gain * tw.x, gain * (h ? -tw.y : tw.y)};
                }
            }
        }
    };
    __syncthreads();
    setup();

    const int t_end = DS * (nsweeps - 1) + SKS * (T - 1) + NU;
    const int n_it = (t_end + 2) & ~1;                  // barrier intervals (t = -1 .. n_it - 2), an even number of them
    const int frame_period = DS * per + SKS;
    int next_need = (loaded - (Q - 1)) * frame_period;
    // bring in the next frame three intervals before its first sweep starts (a tap wave prefetches for the step after next)
    auto load_frames = [&](int t) {
        while (loaded < T + Q - 1 && next_need <= t + 4) {
            const int slot = (loaded % NWR) * NPS;
            const bool evict = loaded >= NWR;
            for (int i = tid; i < Npu; i += nthr) {
                if (evict) gS[(size_t)(loaded - NWR) * Npu + i] = S[slot + dL + i];
                S[slot + dL + i] = gS[(size_t)loaded * Npu + i];
                if constexpr (!BIG) A[slot + dL + i] = gA[(size_t)loaded * Npu + i];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the projection wave's barrier wait is a counted one)
            ++loaded;
            next_need += frame_period;
        }
    };
    // rows of the weights' twiddles used by the two bins of a step: (2 PH, 2 PH + 1) mod Q, plus 4 for every other pair when Q = 8
    auto tw_of = [&](auto ph_c, int u_even, v2f &twa, v2f &twb) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int R0 = (2 * PH) & (Q - 1);
        twa = twg[R0]; twb = twg[(R0 + 1) & (Q - 1)];
        if constexpr (Q == 8) {
            const bool hi = (u_even & 2) != 0;
            twa = hi ? twg[R0 + 4] : twa;
            twb = hi ? twg[R0 + 5] : twb;
        }
    };

    // ---- neighbour-frame tap waves.  KIND 0: frames rho-+r, r >= 2, and rho+1; 1: frame rho-1, whose window ends in a column
    // that is being written (left out, the projection wave adds it) and whose column before that is final only now.
    auto tap_loop = [&](auto kind_c) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int NPRE = KIND == 1 ? NCELL - 3 : NCELL - 2;     // cells that may be read a step early
        v2f wl[2 * NCELL];
        float4 *pw0 = P + (0 * NTW + wave) * 64 + lane, *pw1 = P + (1 * NTW + wave) * 64 + lane;
        int ue = 0 - tstart;                                        // bin pair of this lane at the even step of the pair
        // TWT: cp = (first bin of the pair of steps) mod PT = (2 ue) mod PT; the four bins 2 ue .. 2 ue + 3 are table rows cp .. cp + 3
        auto cp_of = [&](int u_) __attribute__((always_inline)) { const int v = (2 * u_) % PT; return v < 0 ? v + PT : v; };
        int cp = TWT ? cp_of(ue) : 0;
        const int step4 = 4 % PT;
        auto tw_tab = [&](int row, v2f &twa, v2f &twb) __attribute__((always_inline)) {
            const float2 ta = TT[(cp + row) * TQ + r], tb = TT[(cp + row + 1) * TQ + r];
            twa = as_v2f(ta) * gvec; twb = as_v2f(tb) * gvec;
        };
        auto cells = [&](int u_even) __attribute__((always_inline)) {
            const int uc = u_even > -8 ? u_even : -8;               // (a lane far from its start reads in range)
            return reinterpret_cast<const float4 *>(S + fb + 2 * uc);
        };
        auto ld = [&](const float4 *w, auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            const float4 q = w[I];
            wl[2 * I] = (v2f){q.x, q.y}; wl[2 * I + 1] = (v2f){q.z, q.w};
        };
        // sums over this wave's frame X for the bins at window columns C0 + L (a) and C0 + L + 1 (b):
        //   sum_k w[k] X[c-k] + conj(w[k]) X[c+k]
        auto sums = [&](auto c0_c, v2f twa, v2f twb, float4 *pw) __attribute__((always_inline)) {
            constexpr int C0 = decltype(c0_c)::value;
            v2f am = cmul_pk(w0[0], wl[C0 + L]), ap = cmulc_pk(w0[1], wl[C0 + L + 1]);
            v2f bm = cmul_pk(w0[0], wl[C0 + L + 1]), bp = cmulc_pk(w0[1], wl[C0 + L + 2]);
            cmac_pk(am, w0[1], wl[C0 + L - 1]);
            cmac_pk(bm, w0[1], wl[C0 + L]);
#pragma unroll
            for (int k = 2; k <= L; ++k) {
                cmac_pk(am, w0[k], wl[C0 + L - k]);   cmacc_pk(ap, w0[k], wl[C0 + L + k]);
                cmac_pk(bm, w0[k], wl[C0 + L + 1 - k]);
                if (!(KIND == 1 && k == L)) cmacc_pk(bp, w0[k], wl[C0 + L + 1 + k]);
            }
            const v2f pa = cmul_pk(twa, am + ap), pb = cmul_pk(twb, bm + bp);
            *pw = make_float4(pa.x, pa.y, pb.x, pb.y);
        };
        {   // the first pair's early cells
            const float4 *w = cells(ue);
            static_for<NPRE>([&](auto ic) { ld(w, ic); });
        }
        LAB(unsigned long long lab_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lb0 = lab_now();)
        for (int it = 0; it < n_it; it += 2) {
            // (ODD: a sweep may end at either step of a pair -- the lane moves on at the top of the next pair.  Its new sweep
            // starts three steps after the old one ended at the earliest: in this pair's odd step or later; the early cells,
            // read a step later than usual here, are the new window's)
            if constexpr (ODD) {
                if (it > t_done) {
                    s += NSW; setup(); ue = it - tstart;
                    if constexpr (TWT) cp = cp_of(ue);
                    const float4 *w = cells(ue);
                    static_for<NPRE>([&](auto ic) { ld(w, ic); });
                }
            }
            // ---- even step tt = it (interval t = it - 1)
            {
                const float4 *w = cells(ue);
                static_for<NCELL - 1 - NPRE>([&](auto ic) { ld(w, std::integral_constant<int, NPRE + decltype(ic)::value>{}); });
                // frame rho-1 is SKB bins ahead, but the Hermitian images of its bins 4 and 5 (columns 1, 0: this lane's first
                // window) are stored only SKB - 3 bins before this lane starts: final now, not yet when the early cells were read
                if constexpr (KIND == 1) ld(w, std::integral_constant<int, 0>{});
                // A lane that starts at an odd step has its first window one cell further on.  At the smallest odd lag
                // (DS = SKS Q + 1) the frame Q-1 ahead, previous sweep, stores the images of its bins 4 and 5 (cell 1 of that
                // window) in the step the early cells were read in: final now.  (tools/online_schedule_check.py derives
                // which columns of which wave need this, for every Q and lag; tests/test_online_schedule.py runs it.)
                // (only the pair in which such a lane starts: ue = -1)
                if constexpr (ODD && KIND == 0) { if (ue == -1) ld(w, std::integral_constant<int, 1>{}); }
                LAB(const unsigned long long lb1 = lab_now(); lab_acc[0] += lb1 - lb0;)       // late cells of the even step in registers
                v2f twa, twb;
                if constexpr (TWT) tw_tab(0, twa, twb);
                else tw_of(std::integral_constant<int, 0>{}, ue, twa, twb);
                if (!SERIAL) sums(std::integral_constant<int, 0>{}, twa, twb, pw0);
                LAB(lb0 = lab_now(); lab_acc[1] += lb0 - lb1;)                                // sums formed, partial sums stored
            }
            if constexpr (!ODD) { if (it >= t_done) { s += NSW; setup(); ue = it - tstart; if constexpr (TWT) cp = cp_of(ue); } }
            load_frames(it - 1);
            LAB2({ const unsigned long long lbw = lab_now(); lab_acc[2] += lbw - lb0; lb0 = lbw; })   // (level 2) ready for the barrier
            __syncthreads();
            LAB({ const unsigned long long lbb = lab_now(); lab_acc[3] += lbb - lb0; lb0 = lbb; })     // barrier released
            // ---- odd step tt = it + 1
            {
                const float4 *w = cells(ue);
                if constexpr (KIND == 1) ld(w, std::integral_constant<int, NCELL - 2>{});   // its last column is final now
                // ... and for frame rho-1 (SKS steps ahead) an odd start means: the images of its bins 2 .. 5 (cells 2 and 1 of
                // the window) were stored one and two steps after the early cells were read: final only now
                if constexpr (ODD && KIND == 1) { if (ue == -1) { ld(w, std::integral_constant<int, 1>{}); ld(w, std::integral_constant<int, 2>{}); } }
                ld(w, std::integral_constant<int, NCELL - 1>{});
                LAB(const unsigned long long lb1 = lab_now(); lab_acc[0] += lb1 - lb0;)
                v2f twa, twb;
                if constexpr (TWT) tw_tab(2, twa, twb);
                else tw_of(std::integral_constant<int, 1>{}, ue, twa, twb);
                if (!SERIAL) sums(std::integral_constant<int, 2>{}, twa, twb, pw1);
                LAB(lb0 = lab_now(); lab_acc[1] += lb0 - lb1;)
            }
            ue += 2;
            if constexpr (TWT) { cp += step4; cp -= cp >= PT ? PT : 0; }
            load_frames(it);
            asm volatile("" ::: "memory");   // (the partial sums' store stays in front of the reads below)
            {   // the next pair's early cells: columns this wave has used already, under their new names
                // (KIND 1 reads its cell 0 after the barrier anyway -- the compiler would drop an early read of it, and the
                //  count below must be the number of reads that really are issued)
                const float4 *w = cells(ue);
                static_for<NPRE - (KIND == 1 ? 1 : 0)>([&](auto ic) { ld(w, std::integral_constant<int, decltype(ic)::value + (KIND == 1 ? 1 : 0)>{}); });
            }
            LAB2({ const unsigned long long lbw = lab_now(); lab_acc[2] += lbw - lb0; lb0 = lbw; })   // (level 2: the early cells have arrived too)
            // the partial sums must have landed when the projection wave passes the barrier; the early cells need not have (a wave's
            // LDS operations complete in order, and the NPRE youngest are those reads: tools/check_online_isa.py checks the compiled
            // order; a frame load before them drains everything itself, global loads included)
            static_assert(L != 5 || (NPRE == (KIND == 1 ? 4 : 5)), "the counts below are those of L = 5");
            if constexpr (!SERIAL && L == 5 && KIND == 0) asm volatile("s_waitcnt lgkmcnt(5)\n\ts_barrier" ::: "memory");
            else if constexpr (!SERIAL && L == 5 && KIND == 1) asm volatile("s_waitcnt lgkmcnt(3)\n\ts_barrier" ::: "memory");
            else __syncthreads();
            LAB({ const unsigned long long lbb = lab_now(); lab_acc[3] += lbb - lb0; lb0 = lbb; })
        }
        LAB(if (b == 0 && lane == 0) { for (int i = 0; i < 8; ++i) g_lab[8 + hw_wave * 8 + i] = lab_acc[i]; })
    };

    // ---- centre-frame tap wave: its unit's own outputs of this step and the last (columns c-2 .. c, and their Hermitian
    // images near the frame edges) are left out; the window is read afresh every step
    auto centre_loop = [&]() __attribute__((always_inline)) {
        float4 *pw0 = P + (0 * NTW + wave) * 64 + lane, *pw1 = P + (1 * NTW + wave) * 64 + lane;
        int u = 0 - tstart;
        LAB(unsigned long long lab_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lb0 = 0;)
        auto half = [&](auto ph_c, float4 *pw) __attribute__((always_inline)) {
            const int uc = u > -8 ? u : -8;
            const float4 *w = reinterpret_cast<const float4 *>(S + ctb + 2 * uc);
            v2f wl[WN];
#pragma unroll
            for (int i = 0; i < WN / 2; ++i) {
                const float4 q = w[i];
                wl[2 * i] = (v2f){q.x, q.y}; wl[2 * i + 1] = (v2f){q.z, q.w};
            }
            LAB({ const unsigned long long lbr = lab_now(); lab_acc[0] += lbr - lb0; lb0 = lbr; })   // the window is in registers
            // (selects, not branches: some lane of the 64 is near a frame edge in most steps, and a nest of lane-divergent branches
            //  on this wave -- the longest of the step until round 5: profiles/r05_online_phase_budget.json -- cost ~400 clocks a step)
            const int c = 2 * u, g = N - c;
            const v2f zero_ = {0.f, 0.f};
            static_for<(L + 2) / 4>([&](auto iu) {
                constexpr int U = decltype(iu)::value + 1, C = 2 * U;
                const bool at = u == U;
                static_for<3>([&](auto id) {
                    constexpr int D = decltype(id)::value, I = L - 2 * C + D;
                    if constexpr (C - D >= 1 && I >= 0 && I < WN) wl[I] = at ? zero_ : wl[I];
                });
            });
            static_for<(L + 1) / 2 + 1>([&](auto ig) {
                constexpr int G = decltype(ig)::value;
                if constexpr (G & 1) return;                // (N = F - 1 and c are even: g is)
                const bool at = g == G;
                static_for<3>([&](auto id) {
                    constexpr int D = decltype(id)::value, I = L + 2 * G + D;
                    if constexpr (G + D >= 1 && G + D <= L && I < WN) wl[I] = at ? zero_ : wl[I];
                });
            });
            // bin a (column L): taps -1, -2 are columns L-1, L-2 (left out); bin b (column L+1): taps -1 .. -3 are columns L .. L-2
            // (at a frame start those columns are the images of bins 1 and 2 as the previous sweep left them: taken here)
            const bool start = u == 0;
            const v2f zero = {0.f, 0.f}, s1 = start ? wl[L - 1] : zero, s2 = start ? wl[L - 2] : zero;
            v2f ap = cmulc_pk(w0[1], wl[L + 1]), bp = cmulc_pk(w0[1], wl[L + 2]);
            v2f am = cmul_pk(w0[1], s1), bm = cmul_pk(w0[2], s1);
            cmac_pk(am, w0[2], s2);
            if (L >= 3) cmac_pk(bm, w0[L >= 3 ? 3 : 0], s2);
#pragma unroll
            for (int k = 2; k <= L; ++k) {
                if (k >= 3) cmac_pk(am, w0[k], wl[L - k]);
                cmacc_pk(ap, w0[k], wl[L + k]);
                if (k >= 4) cmac_pk(bm, w0[k], wl[L + 1 - k]);
                cmacc_pk(bp, w0[k], wl[L + 1 + k]);
            }
            // (frame offset 0 has no twiddle; whether the centre frame takes part at all is in the weights: setup())
            const v2f pa = am + ap, pb = bm + bp;
            if (!SERIAL) *pw = make_float4(pa.x, pa.y, pb.x, pb.y);
        };
        LAB(lb0 = lab_now();)
        for (int it = 0; it < n_it; it += 2) {
            if constexpr (ODD) { if (it > t_done) { s += NSW; setup(); u = it - tstart; } }
            half(std::integral_constant<int, 0>{}, pw0);
            if constexpr (!ODD) { if (it >= t_done) { s += NSW; setup(); u = it - tstart; } }
            load_frames(it - 1);
            LAB({ const unsigned long long lbw = lab_now(); lab_acc[1] += lbw - lb0; lb0 = lbw; })   // window read, sums formed and stored
            __syncthreads();
            LAB({ const unsigned long long lbb = lab_now(); lab_acc[3] += lbb - lb0; lb0 = lbb; })
            ++u;
            half(std::integral_constant<int, 1>{}, pw1);
            ++u;
            load_frames(it);
            LAB({ const unsigned long long lbw = lab_now(); lab_acc[1] += lbw - lb0; lb0 = lbw; })
            __syncthreads();
            LAB({ const unsigned long long lbb = lab_now(); lab_acc[3] += lbb - lb0; lb0 = lbb; })
        }
        LAB(if (b == 0 && lane == 0) { for (int i = 0; i < 8; ++i) g_lab[8 + hw_wave * 8 + i] = lab_acc[i]; })
    };

    if (hw_wave == Online4Waves<Q>::IDLE) {
        for (int it = 0; it < n_it; it += 2) { load_frames(it - 1); __syncthreads(); load_frames(it); __syncthreads(); }
    } else if (!is_proj) {
        if (wave == 0) centre_loop();
        else if (wave == 1) tap_loop(std::integral_constant<int, 1>{});
        else tap_loop(std::integral_constant<int, 0>{});
    } else {
        // ------------------------------------------------------------------------------------------ projection wave
        // The chain every step waits for, so: as few instructions as possible, and as little latency in the open as possible.
        asm volatile("s_setprio 3");
        if constexpr (SERIAL) {
            for (int t = -1; t < n_it - 1; ++t) {
                const int u = t - tstart;
                if (t >= 0 && valid && u >= 0 && u < NU) {
                    const int c = 2 * u, n = c + L;
                const int e = rho + Q - 1;
                const float2 zero = make_float2(0.f, 0.f);
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int cb = c + bb, nb = n + bb;
                    if (cb >= F) break;
                    const int row = cb % Q, rowneg = (Q - row) % Q;
                    const float2 *wa = W + wset * Q * Q * K1 + row * Q * K1;
                    float2 acc = zero;
                    if (centre) {
                        const float2 *ctr = S + ctb + nb;
#pragma unroll
                        for (int k = 1; k <= L; ++k) pair(acc, wa[k], ctr[-k], ctr[k]);
                    }
#pragma unroll
                    for (int rr = 1; rr < Q; ++rr) {
                        const float2 *lf = S + ((e - rr) % NWR) * NPS + nb;
                        const float2 *rt = S + ((e + rr) % NWR) * NPS + nb;
                        const float2 *wa_r = W + wset * Q * Q * K1 + (row * Q + rr) * K1;
                        const float2 *wb_r = W + wset * Q * Q * K1 + (rowneg * Q + rr) * K1;
                        const bool two = rr < ts;
                        pair(acc, wa_r[0], lf[0], two ? rt[0] : zero);
#pragma unroll
                        for (int k = 1; k <= L; ++k) {
                            pair(acc, wa_r[k], lf[-k], two ? rt[-k] : zero);
                            pair(acc, wb_r[k], two ? rt[k] : zero, lf[k]);
                        }
                    }
                    const int lj = ctb + nb;
                    const float target = A[lj];
                    if (target > thr) {
                        const float mag = sqrtf(acc.x * acc.x + acc.y * acc.y);
                        if (mag > 0.f) {
                            const float2 v = make_float2(acc.x * target / mag, acc.y * target / mag);
                            const float2 vc = make_float2(v.x, -v.y);
                            S[lj] = v;
                            const int nyq = F + L - 1;
                            if (nb >= L + 1 && nb < 2 * L + 1) S[lj + 2 * (L - nb)] = vc;
                            else if (nb >= F - 1 && nb < nyq) S[lj + 2 * (nyq - nb)] = vc;
                        }
                    }
                }
                }
                if (t >= t_done) { s += NSW; setup(); }
                load_frames(t);
                __syncthreads();
            }
        } else {
            // Per step: read the tap waves' sums; while they are in flight form the unit's own terms from operands fetched before
            // the barrier; add up (a tree); re-project the two bins; store; fetch the next step's operands; barrier.  The
            // step-table entry (edge-term set, image offsets, "the pair has a second bin") is fetched a step before it is
            // needed, so that no address waits for a load.  (Tried on top of this, measured, dropped: the next step's operands
            // fetched at the start of a step instead of its end, +1.6 ms; the sums of step t+1 read before the barrier, with a
            // counter the tap waves bump to say whether they were complete, +6 ms -- the centre wave shares this wave's SIMD
            // and is rarely done in time; the tap waves sleeping 64-192 clocks after a barrier to let this wave's reads go
            // first, +0.3-1.7 ms; an idle ninth wave instead of the centre wave on this SIMD, +7 ms; a row stride that spreads
            // the window cells of a ds_read_b128 group over all 16 bank slots, +0.8 ms.  Times: DESIGN 4c.)
            LAB(unsigned long long lab_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lp0 = 0;)
            v2f p1 = {0.f, 0.f}, p2 = {0.f, 0.f};   // current values of columns c-1, c-2 of the unit's frame; zero while the lane
                                                    // has no work and at a frame start (the centre wave reads the images there itself)
            auto lds = [&](unsigned off) __attribute__((always_inline)) { return smem + off; };
            v2f oldA = {0.f, 0.f}, oldB = {0.f, 0.f}, xlate = {0.f, 0.f};
            float target_a = 0.f, target_b = 0.f;
            float4 mA1 = make_float4(0.f, 0.f, 0.f, 0.f), mA2 = mA1, mB1 = mA1, mB2 = mA1, mC = mA1, eA = mA1;   // MT entries 0..5 of the step
            unsigned li_b = oS, ia_b = oS, ib_b = oS;
            bool act = false;
            int4 tab1 = make_int4(0, 0, 0, 0), tab2 = make_int4(0, 0, 0, 0);   // table entries of bin pairs u + 1, u + 2
            unsigned li0 = oS, ai0 = oA, xl0 = oS, et0 = oET;     // per-sweep bases of the lane (byte offsets)
            int u = -1 - tstart;                       // bin pair of the step about to run
            const float *ga0 = gA;      // BIG: magnitudes of the lane's frame, bin 0
            float tq_a = 0.f, tq_b = 0.f;   // ... and those of the pair after the one being fetched, in flight
            v2f twl_t = {0.f, 0.f};         // TWT: twiddle tau_1 of the step's second bin (frame rho-1's late column), fetched with the operands
            auto cpp_of = [&](int u_) __attribute__((always_inline)) { const int v = (2 * u_ + 1) % PT; return v < 0 ? v + PT : v; };
            int cpp = 0;                    // ... its table row: (2 u + 1) mod PT of the bin pair u about to be fetched
            const int step2 = 2 % PT;
            auto derive = [&]() __attribute__((always_inline)) {
                li0 = oS + 8u * (unsigned)(ctb + L);
                ai0 = oA + 4u * (unsigned)(ctb + L);
                // (a lane whose slot holds no sweep, or a frame past the last one, still fetches: keep its row inside the buffer)
                if constexpr (BIG) ga0 = gA + (size_t)(e_own < 0 ? 0 : (e_own > Tp - 1 ? Tp - 1 : e_own)) * Npu + a.Lu;
                xl0 = oS + 8u * (unsigned)(fbm1 + 1 + 2 * L);
                et0 = oET + (unsigned)(wset * NST * 96);
            };
            auto clampu = [&](int un) __attribute__((always_inline)) { return un < 0 ? 0 : (un > NU - 1 ? NU - 1 : un); };
            auto fetch_tab = [&](int un) __attribute__((always_inline)) {
                if constexpr (BIG) {            // the table entry of bin pair un, computed (same formulas as where TAB is filled)
                    const int uu = clampu(un), c = 2 * uu, cb = c + 1, g = N - c;
                    const int stype = (uu >= 1 && uu <= NLO) ? uu : (g < NHI ? NLO + 1 + g : 0);
                    const int da = (c >= 1 && c <= L) ? -16 * c : ((c >= N - L && c <= N - 1) ? 16 * (N - c) : 0);
                    const int db = (cb <= L) ? -16 * cb : ((cb >= N - L && cb <= N - 1) ? 16 * (N - cb) : 0);
                    return make_int4(stype * 96, da, db, cb < F ? 0 : (int)0x80000000);
                } else {
                    return *reinterpret_cast<const int4 *>(lds(oTAB + 16u * (unsigned)clampu(un)));
                }
            };
            auto fetch = [&](int un, int4 tab) __attribute__((always_inline)) {   // operands of bin pair un, whose table entry is `tab`
                act = valid && (unsigned)un < (unsigned)NU;
                const unsigned uc = (unsigned)clampu(un);
                li_b = li0 + 16u * uc;
                const float2 *so = reinterpret_cast<const float2 *>(lds(li_b));
                oldA = as_v2f(so[0]); oldB = as_v2f(so[1]);
                if constexpr (BIG) {
                    // the magnitudes of this pair were requested one fetch ago; those of the next pair are requested now (a lane
                    // that changes frames has two idle steps ahead of it: the queue holds the new frame's values when it needs them).
                    // (bin 2 uc + 1 of the last pair is one past the frame's bins: the caller's buffer has a pad column there)
                    target_a = tq_a;
                    target_b = __int_as_float(__float_as_int(tq_b) | tab.w);
                    const unsigned u1 = (unsigned)clampu(un + 1);
                    tq_a = ga0[2 * u1];
                    tq_b = ga0[2 * u1 + 1];
                } else {
                    const float *ta = reinterpret_cast<const float *>(lds(ai0 + 8u * uc));
                    target_a = ta[0];
                    target_b = __int_as_float(__float_as_int(ta[1]) | tab.w);       // (no second bin: a negative target is never above a threshold)
                }
                const float4 *mt = reinterpret_cast<const float4 *>(lds(et0 + (unsigned)tab.x));
                mA1 = mt[0]; mA2 = mt[1]; mB1 = mt[2]; mB2 = mt[3]; mC = mt[4]; eA = mt[5];
                ia_b = li_b + (unsigned)tab.y;
                ib_b = li_b + 8u + (unsigned)tab.z;
                xlate = as_v2f(*reinterpret_cast<const float2 *>(lds(xl0 + 16u * uc)));
                if constexpr (TWT) twl_t = as_v2f(TT[cpp * TQ + 1]);      // (row (2 un + 1) mod PT, kept incrementally: no division on the chain)
            };
            auto step = [&](auto ph_c, int t) __attribute__((always_inline)) {   // PH: parity of t (and of u)
                constexpr int PH = decltype(ph_c)::value;
                // (this wave has no slack either: 64 / 128 / 256 clocks of s_sleep here cost 2.2 / 4.5 / 7.2 ms, round 5)
                const float4 *pp = P + (PH * NTW) * 64 + lane;
                float4 part[NTW];
#pragma unroll
                for (int w = 0; w < NTW; ++w) part[w] = pp[w * 64];
                __builtin_amdgcn_sched_barrier(0);   // (the reads first: the own terms below cover part of their latency)
                LAB(const unsigned long long lp1 = lab_now(); lab_acc[0] += lp1 - lp0;)   // the tap waves' sums are in registers
                // while the sums are on their way this wave has nothing urgent to issue: the centre-frame wave, which shares its SIMD
                // and is the last to reach the barrier, goes first (its window reads went out ~200 clocks late behind the terms below)
                asm volatile("s_setprio 0");
                // the unit's own history (columns c-1, c-2, their images, the image of bin c) and frame rho-1's late column
                v2f twl = TWT ? twl_t : as_v2f(a.tw[(2 * PH + 1) & (Q - 1)]);
                if constexpr (ODD && Q == 4 && !TWT) twl = ((u ^ PH) & 1) ? as_v2f(a.tw[(2 * PH + 3) & 3]) : twl;   // (odd start: u and t differ in parity)
                if constexpr (Q == 8 && !TWT) twl = (u & 2) ? as_v2f(a.tw[(2 * PH + 5) & (Q - 1)]) : twl;
                v2f ownA = mat_mul_pk(mA1, p1), ownB = mat_mul_pk(mB1, p1);
                mat_mac_pk(ownA, mA2, p2);
                mat_mac_pk(ownB, mB2, p2);
                cmac_cv_pk(ownA, (v2f){eA.x, eA.y}, oldA);
                {
                    const v2f x = cmul_pk(wlate, xlate);
                    cmac_pk(ownB, twl, x);
                }
                asm volatile("s_setprio 3");
                // the tap waves' partial sums, as a tree
                v2f sa[NTW], sb[NTW];
#pragma unroll
                for (int w = 0; w < NTW; ++w) { sa[w] = (v2f){part[w].x, part[w].y}; sb[w] = (v2f){part[w].z, part[w].w}; }
#pragma unroll
                for (int n2 = NTW; n2 > 1; n2 = (n2 + 1) / 2)
#pragma unroll
                    for (int w = 0; w < n2 / 2; ++w) { sa[w] += sa[n2 - 1 - w]; sb[w] += sb[n2 - 1 - w]; }
                v2f accA = sa[0] + ownA, accB = sb[0] + ownB;
                v2f newA;
                {
                    const float m2 = accA.x * accA.x + accA.y * accA.y;      // (in range: the weights carry the spectrogram's scale)
                    const float sc = target_a * __frsqrt_rn(m2);
                    const bool upd = target_a > thr && m2 > 0.f;
                    newA = upd ? accA * sc : oldA;
                }
                mat_mac_pk(accB, mC, newA);        // bin c as tap -1 of bin c+1, and its image as the second bin sees it
                v2f newB;
                {
                    const float m2 = accB.x * accB.x + accB.y * accB.y;
                    const float sc = target_b * __frsqrt_rn(m2);
                    const bool upd = target_b > thr && m2 > 0.f;
                    newB = upd ? accB * sc : oldB;
                }
                // Unchanged bins are written back as they were.  Hermitian images in the pad columns (lwslib.cpp:362-367): a bin
                // without an image has offset 0 -- its conjugate goes to the bin's own place FIRST and is overwritten at once
                // (one wave's LDS stores land in order).  A frame's last pair has no second bin: that place is an image column
                // and gets its old value back (it was read after the store of the step before).
                LAB(const unsigned long long lp2 = lab_now(); lab_acc[1] += lp2 - lp1;)    // own terms, tree, both re-projections
                if (act) {
                    *reinterpret_cast<float2 *>(lds(ia_b)) = make_float2(newA.x, -newA.y);
                    *reinterpret_cast<float2 *>(lds(ib_b)) = make_float2(newB.x, -newB.y);
                    float2 *sn = reinterpret_cast<float2 *>(lds(li_b));
                    sn[0] = make_float2(newA.x, newA.y);
                    sn[1] = make_float2(newB.x, newB.y);
                }
                const v2f zero = {0.f, 0.f};
                p2 = act ? newA : zero;
                p1 = act ? newB : zero;
                load_frames(t);
                // (a lane that changes sweeps has at least two steps without work ahead of it -- NSW DS >= SKS LA + NU + 2 in
                // shape4_of -- : the table entries in flight, which still belong to the old sweep, are used for those two only)
                if (t >= t_done) { s += NSW; setup(); derive(); u = t - tstart; if constexpr (TWT) cpp = cpp_of(u); }
                ++u;
                if constexpr (TWT) { cpp += step2; cpp -= cpp >= PT ? PT : 0; }
                // The stores above must have landed when the other waves pass the barrier; the reads below need not have.  A
                // wave's LDS operations complete in order, so "all but the youngest 7" covers the stores as long as at least 7
                // LDS reads follow them (tests/test_online_isa.py checks the compiled order).  Never more than 15 LDS / scalar
                // memory operations in flight: the counter s_waitcnt tests has four bits (with 16 the engine returned stale
                // registers).
                asm volatile("" ::: "memory");
                const int4 tab3 = fetch_tab(u + 2);
                fetch(u, tab1);
                tab1 = tab2;
                tab2 = tab3;
                // (BIG: the step-table entry is computed and the targets come from global memory -- two LDS reads fewer follow the
                // stores, ~8 remain: a count of 5 keeps a margin of three instructions against a reordering by the compiler)
                LAB2(const unsigned long long lp3 = lab_now(); lab_acc[2] += lp3 - lp2;)   // (level 2) stores landed AND the next operands fetched
                if constexpr (BIG) asm volatile("s_waitcnt lgkmcnt(5)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(7)\n\ts_barrier" ::: "memory");
#if defined(LWS_LAB) && LWS_LAB >= 2
                LAB(lp0 = lab_now(); lab_acc[3] += lp0 - lp3;)                             // waiting in the barrier
#else
                LAB(lp0 = lab_now(); lab_acc[3] += lp0 - lp2;)                             // stores, next fetch, counted wait, barrier (and the fetched operands' arrival: the stamp drains)
#endif
            };
            derive();
            if constexpr (TWT) cpp = cpp_of(u);
            fetch(u, fetch_tab(u));
            tab1 = fetch_tab(u + 1);
            tab2 = fetch_tab(u + 2);
            LAB(lp0 = lab_now();)
            for (int it = 0; it < n_it; it += 2) {
                step(std::integral_constant<int, 1>{}, it - 1);
                step(std::integral_constant<int, 0>{}, it);
            }
            LAB(if (b == 0 && lane == 0) { for (int i = 0; i < 8; ++i) g_lab[8 + hw_wave * 8 + i] = lab_acc[i]; g_lab[0] = (unsigned long long)n_it; })
        }
    }
    const int first_row = loaded > NWR ? loaded - NWR : 0;
    for (int e = first_row; e < loaded; ++e) {
        const int slot = (e % NWR) * NPS;
        for (int i = tid; i < Npu; i += nthr) gS[(size_t)e * Npu + i] = S[slot + dL + i];
    }
}

template <int Q, int L, bool SERIAL, int MAXT> hipError_t launch_qt(const OnlineArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};   // one bit per device
    int attr_dev;
    if (lws::attr_needed(attr_set, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online<Q, L, SERIAL, MAXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lws::attr_done(attr_set, attr_dev);
    }
    hipLaunchKernelGGL((k_online<Q, L, SERIAL, MAXT>), dim3(B), dim3(threads), lds, s, a);
    return hipGetLastError();
}
template <int Q, int L, bool SERIAL> hipError_t launch_q(const OnlineArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    return threads <= 512 ? launch_qt<Q, L, SERIAL, 512>(a, B, threads, lds, s) : launch_qt<Q, L, SERIAL, 1024>(a, B, threads, lds, s);
}

struct Shape { int NSW, threads, DS; size_t lds; bool ok; };

Shape shape_of(int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    Shape sh{0, 0, 0, 0, false};
    if (Qp != Q || L != 5 || !(Q == 2 || Q == 4 || Q == 8) || LA < 0 || n_thr < 1 || T < 1) return sh;
    const int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2, DS_MIN = ((SKB * (Q - 1) + L + 3) / 2), Np = F + 2 * L, per = n_thr + 1;
    const int NU = (F + 1) / 2;
    // The lag between sweeps may be anything from the order-exact minimum up: fewer sweeps in flight, fewer lanes, but
    // proportionally more steps.  Steps are what costs (more waves on a SIMD hide each other's latency almost for free), so
    // take the minimum unless the lanes do not fit a workgroup.
    int DS = 0;
    for (int d = DS_MIN; d <= 4 * DS_MIN && DS == 0; ++d) {
        const int nsw = (NU - 1 + SKS * LA) / d + 2;
        if (nsw * (LA + 1) * Q * 2 <= 1024) DS = d;
    }
    if (DS == 0) return sh;
    sh.DS = DS;
    sh.NSW = (NU - 1 + SKS * LA) / DS + 2;                    // > sweeps in flight
    sh.threads = ((sh.NSW * (LA + 1) * Q * 2 + 63) / 64) * 64;   // two lanes per frame pair
    if (sh.threads > 1024) return sh;
    // Frames alive at once.  The frame loaded at the end of step t (newest frame m_new, (DS*per + SKS) m_new <= t + 1)
    // replaces the one NW rows below it, and the oldest sweep still running (of frame m_lo, t <= DS (per m_lo + per - 1)
    // + SKS m_lo + NU - 1) reads down to row m_lo - LA:  m_new - m_lo <= (DS (per-1) + NU) / (DS per + SKS), and the ring
    // must hold that many frames plus the Q - 1 + LA behind m_lo and the new one.
    const int window = (DS * (per - 1) + NU) / (DS * per + SKS) + LA + Q;
    if (window > NW) return sh;
    sh.lds = ((size_t)NW * Np + 2) * 8 + (size_t)NW * Np * 4 + 8 + (size_t)3 * Q * Q * (L + 1) * 8 + (size_t)Q * 8 + (size_t)n_thr * 4;
    if (sh.lds > 160 * 1024) return sh;
    if ((double)DS * T * per + (double)SKS * T + NU > 1.0e9) return sh;   // step counter is an int
    sh.ok = true;
    return sh;
}

// k_online3: 2Q waves, one lane per (sweep slot, frame position)
Shape shape3_of(int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    Shape sh{0, 0, 0, 0, false};
    if (Qp != Q || L != 5 || !(Q == 2 || Q == 4 || Q == 8) || LA < 0 || LA > 63 || n_thr < 1 || T < 1) return sh;
    const int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2, DS_MIN = ((SKB * (Q - 1) + L + 3) / 2), Np = F + 2 * L, per = n_thr + 1;
    const int NU = (F + 1) / 2;
    sh.NSW = 64 / (LA + 1);
    // order-exact lag; the tap waves run a step ahead: 2 DS >= SKB Q + 2; a slot is free again when its sweep is over:
    // NSW DS >= SKS LA + NU
    int DS = DS_MIN;
    if (2 * DS < SKB * Q + 2) DS = (SKB * Q + 3) / 2;
    const int need = (SKS * LA + NU + sh.NSW - 1) / sh.NSW;
    if (DS < need) DS = need;
    sh.DS = DS;
    sh.threads = (Q == 4 ? 9 : 2 * Q) * 64;
    const int window = (DS * (per - 1) + NU + 1) / (DS * per + SKS) + LA + Q;
    if (window > NW) return sh;
    if (F - 1 < 2 * (L + 3)) return sh;   // frame edges (Hermitian image terms) at least a few steps apart
    sh.lds = (size_t)2 * (2 * Q - 1) * 64 * 16 + (192 + 64) * 8 + ((size_t)NW * Np + 2) * 8 + (size_t)NW * Np * 4 + 8 + (size_t)3 * Q * Q * (L + 1) * 8 +
             (size_t)Q * 8 + (size_t)n_thr * 4;
    if (sh.lds > 160 * 1024) return sh;
    if ((double)DS * T * per + (double)SKS * T + NU > 1.0e9) return sh;
    sh.ok = true;
    return sh;
}

// k_online4: 2Q waves, one lane per (sweep slot, frame position), even lag, ring of NWR frames with an even row stride
struct Shape4 { Shape sh; int NWR, NPS; bool big; };
// big: the kernel's BIG variant (target magnitudes and step table not in LDS)
// PT > 0: the table-twiddle variant (Q in 3..8, twiddle period PT bins: (PT + 3) x 32 (Q > 4: 64) bytes of LDS more)
Shape4 shape4_try(int F, int T, int Lu, int Q, int Qp, int LA, int n_thr, bool big, int PT = 0) {
    Shape4 r{{0, 0, 0, 0, false}, 0, 0, big};
    Shape &sh = r.sh;
    // any stencil half-width up to the kernel's: narrower ones run as L = 5 with zero weights for the taps they do not have
    // (OnlineArgs::Lu) -- the same sums, on a schedule that is order-exact for the wider stencil
    if (Qp != Q || Lu < 1 || Lu > 5 || LA < 0 || LA > 63 || n_thr < 1 || T < 1) return r;
    if (PT > 0 ? (big || Q < 2 || Q > 8 || PT > 512) : !(Q == 2 || Q == 4 || Q == 8)) return r;
    const int L = 5;
    const int SKB = 2 * ((L + 3) / 2), SKS = SKB / 2, DS_MIN = ((SKB * (Q - 1) + L + 3) / 2), Np = F + 2 * L, per = n_thr + 1;
    const int NU = (F + 1) / 2;
    sh.NSW = 64 / (LA + 1);
    // order-exact lag with the tap waves a step ahead (2 DS >= SKB Q + 2: SKS Q + 1 steps.  One step less is order-exact too, but
    // frame rho+Q-1 of the previous sweep is then SKS steps ahead of a lane, as frame rho-1 is: its tap wave must read two cells
    // after the barrier instead of one and leave its last column to the projection wave -- built and measured in round 4: +12 %
    // per step for 6 % fewer steps); a slot is free again, with two steps to spare, when its sweep is over (NSW DS >= SKS LA + NU
    // + 2).  Q = 8 with static twiddles and the verification variant want the lag even.  LWS_ONLINE_LAG_PLUS=k adds k steps
    // (comparison runs: schedules of different lags agree bit for bit).
    int DS = DS_MIN;
    if (2 * DS < SKB * Q + 2) DS = (SKB * Q + 3) / 2;
    const int need = (SKS * LA + NU + 2 + sh.NSW - 1) / sh.NSW;
    if (DS < need) DS = need;
    {
        const char *ep = getenv("LWS_ONLINE_LAG_PLUS"), *es = getenv("LWS_ONLINE_SERIAL_TAPS");   // (the verification variant has no odd build)
        if (ep && atoi(ep) > 0 && atoi(ep) <= 64) DS += atoi(ep);
        const bool odd_ok = (PT > 0 || Q == 4 || Q == 2) && !(es && es[0] == '1');
        if (!odd_ok) DS += DS & 1;
    }
    sh.DS = DS;
    sh.threads = Online4Waves<4>::N == 9 && Q == 4 ? 9 * 64 : 2 * Q * 64;
    if (F - 1 < 2 * (L + 3)) return r;
    r.NPS = Np + (Np & 1);
    // (row strides of Np + 2 .. Np + 44 columns measured in round 5, as in round 3: 40.2-40.8 ms against 40.5 -- the window reads'
    //  bank conflicts are not what a step waits for)
    auto lds_of = [&](int nwr) {
        return (size_t)2 * (2 * Q - 1) * 64 * 16 + (224 + 64) * 8 + ((size_t)nwr * r.NPS + 8) * 8 + (big ? 0 : (size_t)nwr * r.NPS * 4) +
               (size_t)3 * Q * Q * (L + 1) * 8 + (size_t)Q * 8 + (size_t)n_thr * 4 + 16 + (big ? 0 : (size_t)NU * 16) + (PT > 0 ? (size_t)(PT + 3) * (Q <= 4 ? 32 : 64) : 0);
    };
    int nwr_max = 16;
    while (nwr_max > 0 && lds_of(nwr_max) > 160 * 1024) --nwr_max;
    // Frames alive at once (see shape_of; frames are brought in three intervals before their first sweep).  Few iterations
    // per frame on a long frame mean many frames in flight: a longer lag between sweeps trades steps for ring space.
    auto window_of = [&](int ds) { return (ds * (per - 1) + NU + 3) / (ds * per + SKS) + LA + Q; };
    while (window_of(DS) > nwr_max && DS < 16 * sh.DS) DS += 2;
    if (window_of(DS) > nwr_max) return r;
    sh.DS = DS;
    const int window = window_of(DS);
    r.NWR = window + 1 <= nwr_max ? window + 1 : window;
    // two workgroups fit a CU (half of its LDS each) without the spare ring frame but not with it: drop it -- batches of more
    // spectrograms than CUs then run two chains per CU side by side (LWS_ONLINE_SPARE_FRAME=1 keeps it: comparison runs)
    {
        const char *ev = getenv("LWS_ONLINE_SPARE_FRAME");
        if (r.NWR == window + 1 && lds_of(window + 1) > 80 * 1024 && lds_of(window) <= 80 * 1024 && !(ev && ev[0] == '1')) r.NWR = window;
    }
    sh.lds = lds_of(r.NWR);
    if ((double)DS * T * per + (double)SKS * T + NU > 1.0e9) return r;
    sh.ok = true;
    return r;
}
Shape4 shape4_of(int F, int T, int Lu, int Q, int Qp, int LA, int n_thr, int PT = 0) {
    if (PT > 0) {
        const char *ev = getenv("LWS_ONLINE_SERIAL_TAPS");   // (no verification variant with table twiddles: generic engine)
        return (ev && ev[0] == '1') ? Shape4{{0, 0, 0, 0, false}, 0, 0, false} : shape4_try(F, T, Lu, Q, Qp, LA, n_thr, false, PT);
    }
    Shape4 r = shape4_try(F, T, Lu, Q, Qp, LA, n_thr, false);
    const char *ev = getenv("LWS_ONLINE_SERIAL_TAPS");   // (the verification variant has no BIG build)
    if (!r.sh.ok && !(ev && ev[0] == '1')) r = shape4_try(F, T, Lu, Q, Qp, LA, n_thr, true);
    return r;
}

// which layout serves a shape: the wave-per-tap-group one unless it needs much more lag between sweeps (few slots: long
// look-ahead) than the lane-group one; LWS_ONLINE_LAYOUT=2 / 3 forces one (tests)
int pick_layout(const Shape &s2, const Shape &s3, const Shape &s4) {
    const char *ev = getenv("LWS_ONLINE_LAYOUT");
    if (ev && ev[0] == '2' && s2.ok) return 2;
    if (ev && ev[0] == '3' && s3.ok) return 3;
    if (ev && ev[0] == '4' && s4.ok) return 4;
    if (s4.ok && (!s2.ok || 2 * s4.DS <= 3 * s2.DS)) return 4;
    if (s3.ok && (!s2.ok || 2 * s3.DS <= 3 * s2.DS)) return 3;
    return s2.ok ? 2 : 0;
}

template <int Q, int L, bool SERIAL, bool BIG, bool TWT, bool ODD> hipError_t launch_4p(const OnlineArgs &a, int B, size_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online4<Q, L, SERIAL, BIG, TWT, ODD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // (per device: not cached)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_online4<Q, L, SERIAL, BIG, TWT, ODD>), dim3(B), dim3(Online4Waves<Q>::N * 64), lds, s, a);
    return hipGetLastError();
}
template <int Q, int L, bool SERIAL, bool BIG = false, bool TWT = false> hipError_t launch_4(const OnlineArgs &a, int B, size_t lds, hipStream_t s) {
    if constexpr (!SERIAL && (TWT || Q == 2 || Q == 4)) {
        if (a.DS & 1) return launch_4p<Q, L, SERIAL, BIG, TWT, true>(a, B, lds, s);
    }
    if (a.DS & 1) return hipErrorInvalidValue;      // (shape4_try gives these builds an even lag)
    return launch_4p<Q, L, SERIAL, BIG, TWT, false>(a, B, lds, s);
}

template <int Q, int L, bool SERIAL> hipError_t launch_3(const OnlineArgs &a, int B, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};   // one bit per device
    int attr_dev;
    if (lws::attr_needed(attr_set, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online3<Q, L, SERIAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lws::attr_done(attr_set, attr_dev);
    }
    hipLaunchKernelGGL((k_online3<Q, L, SERIAL>), dim3(B), dim3(Online3Waves<Q>::N * 64), lds, s, a);
    return hipGetLastError();
}

}  // namespace

// tw_P, tw_s: the common twiddle structure of the three tensors (online_twiddle), 0 if they have none.  Tensors of Qp = N rows
// (general weights) are served through their first Q rows' base weights like summarised ones.
// (LWS_ONLINE_TABLE_TWIDDLES=1, read when the plan is made: the table variant also where the static one would do -- comparison runs)
bool online_static_twiddles(int Q, int tw_P, int tw_s) {
    const char *ev = getenv("LWS_ONLINE_TABLE_TWIDDLES");
    return tw_P == Q && tw_s == 1 && (Q == 2 || Q == 4 || Q == 8) && !(ev && ev[0] == '1');
}
// `table`: the plan's decision (latched when it was made: it uploaded the table or it did not), never re-read from the environment
bool online_lds_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update, int tw_P, int tw_s, bool table) {
    if (update != 2 || tw_P < 1) return false;
    const bool eighth_turns = tw_P == Q && tw_s == 1 && (Q == 2 || Q == 4 || Q == 8);
    if (!table && !eighth_turns) return false;     // (a structure that needs a table the plan does not have, e.g. Q > 8)
    if (!table)
        return shape_of(F, T, L, Q, Q, LA, n_thr).ok || shape3_of(F, T, L, Q, Q, LA, n_thr).ok || shape4_of(F, T, L, Q, Q, LA, n_thr).sh.ok;
    return shape4_of(F, T, L, Q, Q, LA, n_thr, tw_P).sh.ok;
}

// Common twiddle structure of a weight tensor W[Qp][Q][L+1] (complex128 interleaved): W[p][r][k] == W[0][r][k] exp(2 pi j p r s / P)
// for every row p (create_weights, lws.pyx:160-181: s / P = hop / frame in lowest terms; P = Q, s = 1 when the hop divides the frame).
bool weights_twiddle(const double *W, int Q, int Qp, int L, int pmax, int *P_out, int *s_out) {
    // (*P_out = 0: every weight with r >= 1 is zero -- a tensor that fits any twiddle, e.g. W_ai of hop = frame / 2)
    if (!W || Q < 2 || Qp < 1) return false;
    const int K1 = L + 1;
    double scale = 0;
    for (size_t x = 0; x < (size_t)Qp * Q * K1; ++x) scale = std::fmax(scale, std::hypot(W[2 * x], W[2 * x + 1]));
    if (!(scale > 0)) return false;
    auto at = [&](int p, int r, int k, int c) { return W[2 * (((size_t)p * Q + r) * K1 + k) + c]; };
    auto verify = [&](int P, int sg) {
        if (((long long)Qp * sg) % P != 0) return false;   // the rows a kernel reads besides p = bin: p = Qp - bin (modneg, lwslib.cpp:300,408)
        for (int p = 0; p < Qp; ++p)
            for (int r = 0; r < Q; ++r) {
                const double ang = 2.0 * M_PI * (double)(((long long)p * r * sg) % P) / P;
                const double cs = std::cos(ang), sn = std::sin(ang);
                for (int k = 0; k < K1; ++k) {
                    if (r == 0 && k == 0) continue;   // never read by the kernels
                    const double br = at(0, r, k, 0), bi = at(0, r, k, 1);
                    if (std::hypot(at(p, r, k, 0) - (br * cs - bi * sn), at(p, r, k, 1) - (br * sn + bi * cs)) > 1e-9 * scale) return false;
                }
            }
        return true;
    };
    if (Qp == 1) { *P_out = 1; *s_out = 0; return true; }
    // the turn per bin, theta = s / P, from row 1 against row 0 on the largest weight of the first frame offset r that has one:
    // that gives r theta mod 1, i.e. r candidates for theta
    int rb = 0, kb = 0;
    for (int r = 1; r < Q && rb == 0; ++r)
        for (int k = 0; k < K1; ++k)
            if (std::hypot(at(0, r, k, 0), at(0, r, k, 1)) > std::fmax(1e-6 * scale, rb ? std::hypot(at(0, rb, kb, 0), at(0, rb, kb, 1)) : 0.0)) { rb = r; kb = k; }
    if (rb == 0) {   // nothing but the centre frame: the rows must simply repeat row 0
        if (!verify(1, 0)) return false;
        *P_out = 0; *s_out = 0;
        return true;
    }
    const double br = at(0, rb, kb, 0), bi = at(0, rb, kb, 1), wr = at(1, rb, kb, 0), wi = at(1, rb, kb, 1);
    double tr = std::atan2(wi * br - wr * bi, wr * br + wi * bi) / (2.0 * M_PI);   // arg(w / b) in turns = rb theta mod 1
    tr -= std::floor(tr);
    for (int j = 0; j < rb; ++j) {
        const double theta = (tr + j) / rb;
        for (int P = 1; P <= pmax; ++P) {
            const double sp = theta * P, sr = std::round(sp);
            if (std::fabs(sp - sr) > 1e-7) continue;
            const int sg = (int)sr % P;
            if (verify(P, sg)) { *P_out = P; *s_out = sg; return true; }
            break;                                        // (the smallest P of this candidate failed: multiples of it fail too)
        }
    }
    return false;
}
// the table of k_online4<..., TWT>: [P + 3][TQ] float2, TQ = 4 for Q <= 4 else 8, row p: exp(2 pi j p r s / P), r = 0..TQ-1 (host side;
// the plan uploads it)
void online_twiddle_table(int P, int s, int Q, float *out) {
    const int TQ = Q <= 4 ? 4 : 8;
    for (int p = 0; p < P + 3; ++p)
        for (int r = 0; r < TQ; ++r) {
            const double ang = 2.0 * M_PI * (double)(((long long)p * r * s) % P) / P;
            out[(p * TQ + r) * 2] = (float)std::cos(ang);
            out[(p * TQ + r) * 2 + 1] = (float)std::sin(ang);
        }
}

hipError_t launch_online_lds(const GenericArgs<float> &g, int B, int tw_P, int tw_s, const float *tw_table_dev, hipStream_t stream) {
    OnlineArgs a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    for (int i = 0; i < 3; ++i) a.w[i] = g.w[i].w;
    a.twt = reinterpret_cast<const float2 *>(tw_table_dev); a.PT = tw_P;
    if (tw_table_dev) {                            // table twiddles (the plan uploaded one): the fourth layout only
        const Shape4 t4 = shape4_of(g.F, g.T, g.L, g.Q, g.Q, g.LA, g.n_thr, tw_P);
        if (!t4.sh.ok) return hipErrorInvalidValue;
        for (int q = 0; q < 8; ++q) a.tw[q] = make_float2(1.f, 0.f);   // (unused)
        a.F = g.F; a.T = g.T; a.n_thr = g.n_thr; a.LA = g.LA; a.NSW = t4.sh.NSW; a.DS = t4.sh.DS;
        a.NWR = t4.NWR; a.NPS = t4.NPS; a.Lu = g.L;
        switch (g.Q) {
        case 2: return launch_4<2, 5, false, false, true>(a, B, t4.sh.lds, stream);
        case 3: return launch_4<3, 5, false, false, true>(a, B, t4.sh.lds, stream);
        case 4: return launch_4<4, 5, false, false, true>(a, B, t4.sh.lds, stream);
        case 5: return launch_4<5, 5, false, false, true>(a, B, t4.sh.lds, stream);
        case 6: return launch_4<6, 5, false, false, true>(a, B, t4.sh.lds, stream);
        case 7: return launch_4<7, 5, false, false, true>(a, B, t4.sh.lds, stream);
        default: return launch_4<8, 5, false, false, true>(a, B, t4.sh.lds, stream);
        }
    }
    const Shape sh2 = shape_of(g.F, g.T, g.L, g.Q, g.Q, g.LA, g.n_thr), sh3 = shape3_of(g.F, g.T, g.L, g.Q, g.Q, g.LA, g.n_thr);
    const Shape4 sh4 = shape4_of(g.F, g.T, g.L, g.Q, g.Q, g.LA, g.n_thr);
    const int layout = pick_layout(sh2, sh3, sh4.sh);
    if (layout == 0) return hipErrorInvalidValue;
    const Shape sh = layout == 4 ? sh4.sh : (layout == 3 ? sh3 : sh2);
    for (int q = 0; q < 8; ++q) {
        const double ang = 2.0 * M_PI * q / g.Q;
        // (exact zeros and ones at the quarter turns: the products with them must not pick up rounding)
        double cr = std::cos(ang), sr = std::sin(ang);
        if (std::fabs(cr) < 1e-15) cr = 0;
        if (std::fabs(sr) < 1e-15) sr = 0;
        a.tw[q] = make_float2((float)cr, (float)sr);
    }
    a.F = g.F; a.T = g.T; a.n_thr = g.n_thr; a.LA = g.LA; a.NSW = sh.NSW; a.DS = sh.DS;
    a.NWR = sh4.NWR; a.NPS = sh4.NPS; a.Lu = g.L;
    const char *ev = getenv("LWS_ONLINE_SERIAL_TAPS");   // verification only, see k_online
    if (layout == 4) {
        const bool serial = ev && ev[0] == '1';
        if (sh4.big) {
            if (serial) return hipErrorInvalidValue;
            if (g.Q == 4) return launch_4<4, 5, false, true>(a, B, sh.lds, stream);
            if (g.Q == 2) return launch_4<2, 5, false, true>(a, B, sh.lds, stream);
            return launch_4<8, 5, false, true>(a, B, sh.lds, stream);
        }
        if (g.Q == 4) return serial ? launch_4<4, 5, true>(a, B, sh.lds, stream) : launch_4<4, 5, false>(a, B, sh.lds, stream);
        if (g.Q == 2) return serial ? launch_4<2, 5, true>(a, B, sh.lds, stream) : launch_4<2, 5, false>(a, B, sh.lds, stream);
        return serial ? launch_4<8, 5, true>(a, B, sh.lds, stream) : launch_4<8, 5, false>(a, B, sh.lds, stream);
    }
    if (layout == 3) {
        const bool serial = ev && ev[0] == '1';
        if (g.Q == 4) return serial ? launch_3<4, 5, true>(a, B, sh.lds, stream) : launch_3<4, 5, false>(a, B, sh.lds, stream);
        if (g.Q == 2) return serial ? launch_3<2, 5, true>(a, B, sh.lds, stream) : launch_3<2, 5, false>(a, B, sh.lds, stream);
        return serial ? launch_3<8, 5, true>(a, B, sh.lds, stream) : launch_3<8, 5, false>(a, B, sh.lds, stream);
    }
    if (ev && ev[0] == '1') {
        if (g.Q == 4) return launch_q<4, 5, true>(a, B, sh.threads, sh.lds, stream);
        if (g.Q == 2) return launch_q<2, 5, true>(a, B, sh.threads, sh.lds, stream);
        return launch_q<8, 5, true>(a, B, sh.threads, sh.lds, stream);
    }
    if (g.Q == 4) return launch_q<4, 5, false>(a, B, sh.threads, sh.lds, stream);
    if (g.Q == 2) return launch_q<2, 5, false>(a, B, sh.threads, sh.lds, stream);
    return launch_q<8, 5, false>(a, B, sh.threads, sh.lds, stream);
}

}  // namespace lws
