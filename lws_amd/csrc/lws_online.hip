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

namespace lws {
namespace {

constexpr int NW = 16;  // frames in the LDS ring

struct OnlineArgs {
    float2 *state;       // [B][Tp][Np]
    const float *amp;    // [B][Tp][Np]
    const float *thr;    // [B][n_thr]
    const float2 *w[3];  // W, W_ai, W_af: [Q][Q][L+1], zero where flagged off
    float2 tw[8];        // exp(2 pi j q / Q), q < Q
    int F, T, n_thr, LA, NSW;
    int DS;              // steps between consecutive sweeps (>= the order-exact minimum, see shape_of)
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

template <int Q, int L, bool SERIAL, int MAXT> hipError_t launch_qt(const OnlineArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_online<Q, L, SERIAL, MAXT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_online<Q, L, SERIAL, MAXT>), dim3(B), dim3(threads), lds, s, a);
    return hipGetLastError();
}
template <int Q, int L, bool SERIAL> hipError_t launch_q(const OnlineArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    return threads <= 512 ? launch_qt<Q, L, SERIAL, 512>(a, B, threads, lds, s) : launch_qt<Q, L, SERIAL, 1024>(a, B, threads, lds, s);
}

struct Shape { int NSW, threads, DS; size_t lds; bool ok; };

Shape shape_of(int F, int T, int L, int Q, int Qp, int LA, int n_thr) {
    Shape sh{0, 0, 0, false};
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

}  // namespace

bool online_lds_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update, bool twiddle_structure) {
    return update == 2 && twiddle_structure && shape_of(F, T, L, Q, Qp, LA, n_thr).ok;
}

// W[p][r][k] == W[0][r][k] exp(2 pi j p r / Q) for every p, r, k (what create_weights produces, lws.pyx:160-181)
bool weights_have_twiddle_structure(const double *W, int Q, int Qp, int L) {
    if (!W || Qp != Q) return false;
    const int K1 = L + 1;
    double scale = 0;
    for (int x = 0; x < Q * Q * K1; ++x) scale = std::fmax(scale, std::hypot(W[2 * x], W[2 * x + 1]));
    for (int p = 0; p < Q; ++p)
        for (int r = 0; r < Q; ++r)
            for (int k = 0; k <= L; ++k) {
                if (r == 0 && k == 0) continue;  // never read
                const double ang = 2.0 * M_PI * p * r / Q;
                const double br = W[2 * (r * K1 + k)], bi = W[2 * (r * K1 + k) + 1];
                const double er = br * std::cos(ang) - bi * std::sin(ang), ei = br * std::sin(ang) + bi * std::cos(ang);
                const double wr = W[2 * ((p * Q + r) * K1 + k)], wi = W[2 * ((p * Q + r) * K1 + k) + 1];
                if (std::hypot(wr - er, wi - ei) > 1e-9 * scale) return false;
            }
    return true;
}

hipError_t launch_online_lds(const GenericArgs<float> &g, int B, hipStream_t stream) {
    const Shape sh = shape_of(g.F, g.T, g.L, g.Q, g.Qp, g.LA, g.n_thr);
    if (!sh.ok) return hipErrorInvalidValue;
    OnlineArgs a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    for (int i = 0; i < 3; ++i) a.w[i] = g.w[i].w;
    for (int q = 0; q < 8; ++q) {
        const double ang = 2.0 * M_PI * q / g.Q;
        // (exact zeros and ones at the quarter turns: the products with them must not pick up rounding)
        double cr = std::cos(ang), sr = std::sin(ang);
        if (std::fabs(cr) < 1e-15) cr = 0;
        if (std::fabs(sr) < 1e-15) sr = 0;
        a.tw[q] = make_float2((float)cr, (float)sr);
    }
    a.F = g.F; a.T = g.T; a.n_thr = g.n_thr; a.LA = g.LA; a.NSW = sh.NSW; a.DS = sh.DS;
    const char *ev = getenv("LWS_ONLINE_SERIAL_TAPS");   // verification only, see k_online
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
