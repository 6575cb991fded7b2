// lws_systolic.hip -- the fast path for batch LWS (LWSQ2 / LWSQ4 / LWSanyQ, lwslib.cpp:72-373) on
// gfx950: an order-exact *systolic* re-statement of the in-place Gauss-Seidel sweep.
//
// One workgroup (8 compute waves + 1 service wave) owns one spectrogram and keeps I = 8 consecutive
// sweeps in flight.  A lane is a (sweep, frame) processor that marches along the bins of its frame,
// one bin per step; the 64 lanes of compute wave i work on 64 consecutive frames of sweep
// "iteration g*I + i", frame m trailing frame m-1 by SKEW = 8 bins, and sweep j+1 trailing sweep j
// by LAG = 32 steps:
//
//      bin (frame m, bin c) of the sweep handled by wave i runs at step  t = 8*m + c + 32*(i+1) (+ group offset)
//
// SKEW > L and LAG > L + (Q-1)*SKEW guarantee that every "new" neighbour -- (m, c-k), (m-r, c+-k) --
// has already been produced and every "old" neighbour -- (m, c+k), (m+r, c+-k) -- has not yet been
// overwritten by this sweep but has been finished by the previous one: exactly the values the
// sequential reference reads (same argument as the skewed wavefront of lws_generic.hip; SURVEY.md
// facts 1 and 12).  A produced value is consumed by its 7 consumer lanes within 31 steps, so the
// whole exchange runs through per-wave rings in LDS:
//
//      ring[set][step mod 32][lane]  (float2)      set s = output of compute wave s-1, set 0 = loader
//
// i.e. a value is addressed by WHEN it was produced, not by where it lives in the spectrogram, and a
// reader's address is  lane_base + ((t + const) mod 32)*512  with a compile-time `const` per stencil
// tap (the step loop is unrolled by 8 so that the bin phase -- and with it bin % Q, the twiddle of
// the weights and the ring slot -- is static).  HBM is touched once per 8 sweeps: the service wave
// streams the spectrogram in (set 0) ahead of wave 0, wave 7 streams it out, both through a
// time-skewed global layout  state_w[(8m + c) mod G][m mod 64]  in which every access of a wave is
// 64 consecutive elements.
//
// Frequency edges: Hermitian images below DC / above Nyquist are never stored; the one lane of a wave
// that is within L bins of a frame edge re-reads the image taps from the mirrored bin (conjugated)
// under its own exec mask.  The Nyquist bin (bin F-1) does not fit the 512-step frame period and is
// computed by the service wave (one lane per sweep in flight), which also runs the loader.
//
// Scope of this kernel: summarised weights with the twiddle structure create_weights produces
// (lws.pyx:160-181: W[p][r][k] = W[0][r][k]*exp(2j*pi*p*r/Q)), Q in {2,4}, L <= 7 and (Q-1)*8+L+1 <= 32,
// F-1 a multiple of 8 and <= 512, fp32.  Anything else is served by the generic engine.
#include "lws_systolic.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>

namespace lws {
namespace {

constexpr int LANES = 64;
constexpr int RING = 32;
constexpr int SLOT_BYTES = LANES * 8;                    // one ring slot: 64 float2
constexpr int SET_BYTES = RING * SLOT_BYTES;             // 16 KiB
constexpr int NSLOTS = 8;                                // sweeps in flight (compute waves)
constexpr int NSETS = NSLOTS + 1;
constexpr int NYQ_OFF = NSETS * SET_BYTES;               // Nyquist values: [set][lane] float2
constexpr int THR_OFF = NYQ_OFF + NSETS * SLOT_BYTES;    // effective thresholds: floats
constexpr int MAX_ITERS = 440;
constexpr int META_OFF = THR_OFF + MAX_ITERS * 4;        // n_eff
constexpr int LDS_BYTES = META_OFF + 16;
constexpr int SKEW = 8, ROWP = SKEW * LANES, LAG = 32, PF = 8;
constexpr int NTHREADS = LANES * (NSLOTS + 1);
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// Where a stencil tap is found.  off: production time relative to the reader's clock (already
// includes -LAG for "old" values); set_new: ring set of the reader's own sweep (1) or of the
// previous sweep (0).
enum { K_RING = 0, K_NYQ = 1, K_SELF = 2 };
struct Src { int kind, set_new, off, conj; };

__host__ __device__ constexpr Src src_normal(int dr, int dk) {
    const bool is_new = dr < 0 || (dr == 0 && dk < 0);
    return Src{K_RING, is_new ? 1 : 0, SKEW * dr + dk - (is_new ? 0 : LAG), 0};
}
// reader at bin c = P (first bins of a frame), tap at c + dk < 0: image of bin -(c+dk), conjugated
__host__ __device__ constexpr Src src_start(int P, int dr, int dk) {
    const int cm = -(P + dk);            // mirrored bin, 1..L
    const int off0 = SKEW * dr + cm - P;
    if (dr == 0 && cm == P) return Src{K_SELF, 0, 0, 1};
    const bool is_new = dr < 0 || (dr == 0 && cm < P);
    return Src{K_RING, is_new ? 1 : 0, off0 - (is_new ? 0 : LAG), 1};
}
// reader at bin c = C - 8 + P (last bins of a frame), tap at c + dk >= C: the Nyquist bin (dk == e) or
// the image of bin 2C - (c+dk), conjugated; e = C - c = 8 - P
__host__ __device__ constexpr Src src_end(int P, int dr, int dk) {
    const int e = 8 - P;
    if (dk == e) return Src{K_NYQ, dr < 0 ? 1 : 0, 0, 0};
    const int d = 2 * e - dk;            // mirrored bin minus reader bin
    if (dr == 0 && d == 0) return Src{K_SELF, 0, 0, 1};
    const bool is_new = dr < 0 || (dr == 0 && d < 0);
    return Src{K_RING, is_new ? 1 : 0, SKEW * dr + d - (is_new ? 0 : LAG), 1};
}

struct SysArgs {
    float2 *state_w;         // [B][G][64]   time-skewed spectrogram
    const float *amp_w;      // [B][G][64]
    float2 *state_nyq;       // [B][TpPad]
    const float *amp_nyq;    // [B][TpPad]
    const float *thr;        // [B][n_iters] thresholds scaled by mean|S|
    const float *amax;       // [B] max target magnitude
    int n_iters, T, Tp, TpPad, Kr, G, C;
    float w[2 * 4 * 8];      // W[0][r][k] as (re, im), r < Q, k <= L (at most 4 x 8)
};

__device__ __forceinline__ float2 lds_read(int addr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    return *reinterpret_cast<const float2 *>(smem + addr);
}
__device__ __forceinline__ void lds_write(int addr, float2 v) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    *reinterpret_cast<float2 *>(smem + addr) = v;
}
__device__ __forceinline__ float2 cj(float2 v) { return make_float2(v.x, -v.y); }

// Loads that must observe what another wave of this workgroup stored earlier: bypass the per-CU L1.
__device__ __forceinline__ float2 load_l2(const float2 *p) {
    const unsigned long long u =
        __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float2 v;
    v.x = __uint_as_float((unsigned)(u & 0xffffffffull));
    v.y = __uint_as_float((unsigned)(u >> 32));
    return v;
}

// Per-lane registers of a compute lane that stay valid for one block of 8 steps.
struct LaneCtx {
    int nb[4][4];     // [d][m]: LDS address of lane (rho - d) in the own (new) set, block (a - m) & 3
    int ob[4][4];     // [d][m]: lane (rho + d) in the previous sweep's (old) set
    int nyq_n[4], nyq_o[4];
    bool is_start, is_end, live, store;
    float thr;
};

// address of the ring entry produced OFF steps relative to the current clock (phase P of block a)
template <int P, int OFF> __device__ __forceinline__ int ring_addr(const int (&base)[4]) {
    constexpr int q = P + OFF;
    static_assert(q >= -32 && q <= 7, "ring retention exceeded");
    constexpr int fl = (q >= 0) ? 0 : -((-q + 7) / 8);   // floor(q / 8)
    constexpr int m = (-fl) & 3;
    constexpr int within = q - 8 * fl;
    return base[m] + within * SLOT_BYTES;
}

template <int P, int DR, int DK, int EDGE>  // EDGE: 0 normal, 1 frame start, 2 frame end
__device__ __forceinline__ float2 tap(const LaneCtx &cx, float2 self_old) {
    constexpr Src s = (EDGE == 0) ? src_normal(DR, DK) : (EDGE == 1 ? src_start(P, DR, DK) : src_end(P, DR, DK));
    constexpr int d = DR < 0 ? -DR : DR;
    float2 v;
    if constexpr (s.kind == K_SELF) v = self_old;
    else if constexpr (s.kind == K_NYQ) v = lds_read(s.set_new ? cx.nyq_n[d] : cx.nyq_o[d]);
    else {
        static_assert(s.off <= -1 && s.off >= -31, "tap outside ring retention");
        v = lds_read(s.set_new ? ring_addr<P, s.off>(cx.nb[d]) : ring_addr<P, s.off>(cx.ob[d]));
    }
    if constexpr (s.conj) v = cj(v);
    return v;
}

// acc += w*b + conj(w)*c with w = (wr, wi) * j^ROT   (grouped form of lwslib.cpp:310-311)
template <int ROT> __device__ __forceinline__ void pair_rot(float2 &a, float wr, float wi, float2 b, float2 c) {
    const float sx = b.x + c.x, dy = b.y - c.y, sy = b.y + c.y, dx = b.x - c.x;
    if constexpr (ROT == 0) { a.x += wr * sx - wi * dy; a.y += wr * sy + wi * dx; }
    else if constexpr (ROT == 1) { a.x += -wi * sx - wr * dy; a.y += -wi * sy + wr * dx; }
    else if constexpr (ROT == 2) { a.x += -wr * sx + wi * dy; a.y += -wr * sy - wi * dx; }
    else { a.x += wi * sx + wr * dy; a.y += wi * sy - wr * dx; }
}

// The weighted sum of one bin at phase P (bin % 8 == P), all taps with compile-time ring offsets.
template <int Q, int L, uint32_t MASK, int P>
__device__ __forceinline__ float2 weighted_sum(const SysArgs &a, const LaneCtx &cx, float2 self_old) {
    constexpr int K1 = L + 1;
    float2 acc = make_float2(0.f, 0.f);
    // centre frame: W[.,0,k] does not depend on bin % Q
    {
        float2 lo[L], hi[L];
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> k) & 1u) {
                lo[k - 1] = tap<P, 0, -k, 0>(cx, self_old);
                hi[k - 1] = tap<P, 0, k, 0>(cx, self_old);
            }
        });
        if constexpr (P < L) {
            if (cx.is_start)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> k) & 1u) && (P - k < 0)) lo[k - 1] = tap<P, 0, -k, 1>(cx, self_old);
                });
        }
        if constexpr (P + L >= 8) {
            if (cx.is_end)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> k) & 1u) && (P + k >= 8)) hi[k - 1] = tap<P, 0, k, 2>(cx, self_old);
                });
        }
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> k) & 1u) pair_rot<0>(acc, a.w[2 * k], a.w[2 * k + 1], lo[k - 1], hi[k - 1]);
        });
    }
    static_for<Q - 1>([&](auto ir) {
        constexpr int r = decltype(ir)::value + 1;
        constexpr int mod = P % Q;
        constexpr int rot = ((mod * r) % Q) * (4 / Q);  // quarter turns of exp(2j*pi*mod*r/Q)
        // taps of frames m-r (up) and m+r (dn), bins c-L .. c+L
        float2 up[2 * L + 1], dn[2 * L + 1];
        static_for<2 * L + 1>([&](auto id) {
            constexpr int dk = decltype(id)::value - L;
            constexpr int k = dk < 0 ? -dk : dk;
            if constexpr ((MASK >> (r * K1 + k)) & 1u) {
                up[dk + L] = tap<P, -r, dk, 0>(cx, self_old);
                dn[dk + L] = tap<P, r, dk, 0>(cx, self_old);
            }
        });
        if constexpr (P < L) {
            if (cx.is_start)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> (r * K1 + k)) & 1u) && (P - k < 0)) {
                        up[L - k] = tap<P, -r, -k, 1>(cx, self_old);
                        dn[L - k] = tap<P, r, -k, 1>(cx, self_old);
                    }
                });
        }
        if constexpr (P + L >= 8) {
            if (cx.is_end)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> (r * K1 + k)) & 1u) && (P + k >= 8)) {
                        up[L + k] = tap<P, -r, k, 2>(cx, self_old);
                        dn[L + k] = tap<P, r, k, 2>(cx, self_old);
                    }
                });
        }
        if constexpr ((MASK >> (r * K1)) & 1u)
            pair_rot<rot>(acc, a.w[2 * (r * K1)], a.w[2 * (r * K1) + 1], up[L], dn[L]);
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> (r * K1 + k)) & 1u) {
                // W[mod]*S[m-r,c-k] + conj(W[mod])*S[m+r,c-k] + W[-mod]*S[m+r,c+k] + conj(W[-mod])*S[m-r,c+k]
                // with W[-mod] = +-W[mod] for a real / imaginary twiddle (the LWSQ2 / LWSQ4 grouping)
                const float wr = a.w[2 * (r * K1 + k)], wi = a.w[2 * (r * K1 + k) + 1];
                float2 b, c;
                if constexpr ((rot & 1) == 0) {
                    b = make_float2(up[L - k].x + dn[L + k].x, up[L - k].y + dn[L + k].y);
                    c = make_float2(dn[L - k].x + up[L + k].x, dn[L - k].y + up[L + k].y);
                } else {
                    b = make_float2(up[L - k].x - dn[L + k].x, up[L - k].y - dn[L + k].y);
                    c = make_float2(dn[L - k].x - up[L + k].x, dn[L - k].y - up[L + k].y);
                }
                pair_rot<rot>(acc, wr, wi, b, c);
            }
        });
    });
    return acc;
}

// Magnitude re-projection (lwslib.cpp:356-360): keep the old value unless the bin is active and the sum is non-zero.
__device__ __forceinline__ float2 project(float2 acc, float target, bool active, float2 old) {
    const float mag = sqrtf(acc.x * acc.x + acc.y * acc.y);
    const bool ok = active && (mag > 0.f);
    const float s = target / mag;
    return ok ? make_float2(acc.x * s, acc.y * s) : old;
}

// Per-lane bookkeeping of a sweep processor at clock v: which (sweep, frame) it is working on.
struct RowInfo { bool valid, real; int j, me; };
__device__ __forceinline__ RowInfo row_info(int vv /* v - 8*rho, start of frame relative clock */, int slot,
                                            const SysArgs &a, int n_eff, int Q) {
    RowInfo r;
    const int kap = vv >> 9;  // ROWP == 512
    const int g = kap / a.Kr, k = kap - g * a.Kr;
    r.me = k * LANES;  // caller adds rho
    r.j = g * NSLOTS + slot;
    r.valid = (vv >= 0) && (r.j < n_eff);
    r.real = false;
    (void)Q;
    return r;
}

template <int Q, int L, uint32_t MASK, int P>
__device__ __forceinline__ void compute_step(const SysArgs &a, const LaneCtx &cx, int lane, int v, float2 &self_old,
                                             float (&ampq)[PF], float2 *state_w_b, const float *amp_w_b, int G) {
    // taps, weighted sum, projection
    const float2 acc = weighted_sum<Q, L, MASK, P>(a, cx, self_old);
    const float target = ampq[P];
    const bool active = cx.live && (target > cx.thr);
    const float2 out = project(acc, target, active, self_old);
    // publish: own set, slot (v mod 32) = block m = 0, within = P
    lds_write(cx.nb[0][0] + P * SLOT_BYTES, out);
    if (cx.store) state_w_b[(size_t)(v % G) * LANES + lane] = out;
    // prefetch for later steps: own old value of the next step (age 31 now), target magnitude 8 steps ahead
    self_old = lds_read(ring_addr<P, -31>(cx.ob[0]));
    ampq[P] = amp_w_b[(size_t)((v + PF + G) % G) * LANES + lane];  // + G: clocks start negative
}

template <int Q, int L, uint32_t MASK>
__global__ void __launch_bounds__(NTHREADS, 3) k_systolic(SysArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *thr_eff = reinterpret_cast<float *>(smem + THR_OFF);
    int *meta = reinterpret_cast<int *>(smem + META_OFF);
    const int G = a.G, C = a.C, Kr = a.Kr;
    float2 *state_w_b = a.state_w + (size_t)b * G * LANES;
    const float *amp_w_b = a.amp_w + (size_t)b * G * LANES;
    float2 *state_nyq_b = a.state_nyq + (size_t)b * a.TpPad;
    const float *amp_nyq_b = a.amp_nyq + (size_t)b * a.TpPad;

    // sweeps whose threshold is not below the largest magnitude cannot change anything: drop them
    if (threadIdx.x == 0) {
        const float amax = a.amax[b];
        int n = 0;
        for (int i = 0; i < a.n_iters; ++i) {
            const float th = a.thr[(size_t)b * a.n_iters + i];
            if (amax > th) thr_eff[n++] = th;
        }
        meta[0] = n;
    }
    // poison-free start: rings may hold anything, but zero keeps the arithmetic of idle lanes finite
    for (int i = threadIdx.x; i < THR_OFF / 8; i += NTHREADS) reinterpret_cast<float2 *>(smem)[i] = make_float2(0.f, 0.f);
    __syncthreads();
    const int n_eff = meta[0];
    if (n_eff == 0) return;
    const int n_groups = (n_eff + NSLOTS - 1) / NSLOTS;
    // slot i runs on clock v_i = t - (i+1)*LAG; the loader (virtual slot -1) on clock t
    const int t_end = (n_groups - 1) * G + (NSLOTS + 1) * LAG + SKEW * a.Tp + ROWP + 8;

    if (wave < NSLOTS) {
        // ------------------------------------------------------------------ compute wave = sweep slot
        const int slot = wave;
        LaneCtx cx;
        float2 self_old = make_float2(0.f, 0.f);
        float ampq[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) ampq[i] = 0.f;
        const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
        for (int t0 = 0; t0 < t_end; t0 += 8) {
            const int v0 = t0 - (slot + 1) * LAG;  // clock at phase 0 of this block (multiple of 8)
            // ---- block prologue: where is this lane?
            const int vv = v0 - SKEW * lane;        // clock relative to the start of lane's first frame
            const int cbase = vv & (ROWP - 1);
            const int kap = vv >> 9;
            const int g = kap / Kr, k = kap - g * Kr;
            const int me = k * LANES + lane;
            const int j = g * NSLOTS + slot;
            const bool valid = (vv >= 0) && (j < n_eff) && (me < a.Tp);
            const bool real_row = valid && (me >= Q - 1) && (me < a.T + Q - 1);
            cx.live = real_row && (cbase < C);
            cx.store = valid && (cbase < C) && (slot == NSLOTS - 1 || j == n_eff - 1);
            cx.is_start = (cbase == 0);
            cx.is_end = (cbase == C - 8);
            cx.thr = thr_eff[(valid ? j : 0)];
            const int ablk = (v0 >> 3);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int ln = ((lane - d) & 63) * 8, lo = ((lane + d) & 63) * 8;
                cx.nyq_n[d] = NYQ_OFF + (slot + 1) * SLOT_BYTES + ln;
                cx.nyq_o[d] = NYQ_OFF + slot * SLOT_BYTES + lo;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int blk = ((ablk - m) & 3) * (8 * SLOT_BYTES);
                    cx.nb[d][m] = set_new + blk + ln;
                    cx.ob[d][m] = set_old + blk + lo;
                }
            }
            // one block early, so that the prefetched own-old value and magnitudes are warm at the first bin
            const bool any_valid = __any((vv >= -8) && (j < n_eff || vv < 0));
            // ---- 8 steps, phase static
            static_for<8>([&](auto ip) {
                constexpr int P = decltype(ip)::value;
                if (any_valid)
                    compute_step<Q, L, MASK, P>(a, cx, lane, v0 + P, self_old, ampq, state_w_b, amp_w_b, G);
                __syncthreads();
            });
        }
    } else {
        // ------------------------------------------------------------------ service wave: loader + Nyquist bins
        float2 pend[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) pend[i] = load_l2(state_w_b + (size_t)(i % G) * LANES + lane);  // clocks 0..7
        // Nyquist lanes: lane l < NSLOTS serves slot l; lane NSLOTS loads Nyquist values for set 0
        float nyq_amp_next = 0.f;
        float2 nyq_in_next = make_float2(0.f, 0.f);
        constexpr int K1 = L + 1;
        for (int t0 = 0; t0 < t_end; t0 += 8) {
            const int ablk = (t0 >> 3);
            // ---- Nyquist bins fall on phase 0 (C is a multiple of 8)
            {
                const int slot = lane;                       // lanes 0..7
                const bool is_nyq_lane = lane < NSLOTS;
                const bool is_nyq_loader = lane == NSLOTS;
                const int v0 = t0 - (is_nyq_lane ? (slot + 1) * LAG : 0);
                const int vrow = (v0 - C) >> 3;              // virtual frame whose Nyquist bin is due now
                const int rho = vrow & 63, kap = vrow >> 6;
                const int g = kap / Kr, k = kap - g * Kr;
                const int me = k * LANES + rho;
                const int j = g * NSLOTS + (is_nyq_lane ? slot : -1);
                const bool valid = (v0 - C >= 0) && (me < a.Tp) &&
                                   (is_nyq_lane ? (j < n_eff) : (is_nyq_loader && g < n_groups));
                if (is_nyq_loader) {
                    // value loaded one block ago belongs to this frame
                    lds_write(NYQ_OFF + rho * 8, nyq_in_next);
                    // next block's frame
                    const int vr1 = vrow + 1, rho1 = vr1 & 63, kap1 = vr1 >> 6;
                    const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * LANES + rho1;
                    if (vr1 >= 0 && me1 < a.Tp) nyq_in_next = load_l2(state_nyq_b + me1);
                }
                if (is_nyq_lane) {
                    const float target = nyq_amp_next;
                    const bool real_row = valid && (me >= Q - 1) && (me < a.T + Q - 1);
                    const float thr = thr_eff[valid ? j : 0];
                    const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
                    int nb[4][4], ob[4][4], nn[4], no[4];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ln = ((rho - d) & 63) * 8, lo = ((rho + d) & 63) * 8;
                        nn[d] = NYQ_OFF + (slot + 1) * SLOT_BYTES + ln;
                        no[d] = NYQ_OFF + slot * SLOT_BYTES + lo;
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const int blk = ((ablk - m) & 3) * (8 * SLOT_BYTES);
                            nb[d][m] = set_new + blk + ln;
                            ob[d][m] = set_old + blk + lo;
                        }
                    }
                    const float2 old = lds_read(no[0]);
                    float2 acc = make_float2(0.f, 0.f);
                    // bin C: bin % Q == 0, every twiddle is 1; taps above Nyquist are conjugated images
                    static_for<L>([&](auto ik) {
                        constexpr int k = decltype(ik)::value + 1;
                        if constexpr ((MASK >> k) & 1u) {
                            const float2 lo = lds_read(ring_addr<0, -k>(nb[0]));
                            pair_rot<0>(acc, a.w[2 * k], a.w[2 * k + 1], lo, cj(lo));
                        }
                    });
                    static_for<Q - 1>([&](auto ir) {
                        constexpr int r = decltype(ir)::value + 1;
                        if constexpr ((MASK >> (r * K1)) & 1u)
                            pair_rot<0>(acc, a.w[2 * r * K1], a.w[2 * r * K1 + 1], lds_read(nn[r]), lds_read(no[r]));
                        static_for<L>([&](auto ik) {
                            constexpr int k = decltype(ik)::value + 1;
                            if constexpr ((MASK >> (r * K1 + k)) & 1u) {
                                const float2 up = lds_read(ring_addr<0, -SKEW * r - k>(nb[r]));
                                const float2 dn = lds_read(ring_addr<0, SKEW * r - k - LAG>(ob[r]));
                                const float2 bsum = make_float2(up.x + dn.x, up.y - dn.y);   // up + conj(dn)
                                const float2 csum = make_float2(dn.x + up.x, dn.y - up.y);   // dn + conj(up)
                                pair_rot<0>(acc, a.w[2 * (r * K1 + k)], a.w[2 * (r * K1 + k) + 1], bsum, csum);
                            }
                        });
                    });
                    const bool active = real_row && (target > thr);
                    const float2 out = project(acc, target, active, old);
                    lds_write(nn[0], out);
                    if (valid && (slot == NSLOTS - 1 || j == n_eff - 1)) state_nyq_b[me] = out;
                    // target magnitude of the next block's Nyquist bin
                    const int vr1 = vrow + 1, rho1 = vr1 & 63, kap1 = vr1 >> 6;
                    const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * LANES + rho1;
                    nyq_amp_next = (vr1 >= 0 && me1 < a.Tp) ? amp_nyq_b[me1] : 0.f;
                }
            }
            // ---- loader: feed set 0 with the values the virtual previous sweep would produce, 8 steps ahead
            static_for<8>([&](auto ip) {
                constexpr int P = decltype(ip)::value;
                const int v = t0 + P;
                lds_write(((ablk & 3) * 8 + P) * SLOT_BYTES + lane * 8, pend[P]);
                pend[P] = load_l2(state_w_b + (size_t)((v + PF) % G) * LANES + lane);
                __syncthreads();
            });
        }
    }
}

// ---------------------------------------------------------------------------------------------
// layout conversion: reference extended layout <-> time-skewed layout
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_to_skew(const float2 *state, const float *amp, float2 *state_w, float *amp_w,
                                                  float2 *state_nyq, float *amp_nyq, unsigned *amax_bits, int T, int F,
                                                  int L, int Q, int G, int TpPad) {
    const int me = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), C = F - 1;
    const float2 *srow = state + ((size_t)b * Tp + me) * Np + L;
    const float *arow = amp + ((size_t)b * Tp + me) * Np + L;
    float2 *sw = state_w + (size_t)b * G * LANES;
    float *aw = amp_w + (size_t)b * G * LANES;
    const bool real_row = me >= Q - 1 && me < T + Q - 1;
    float mx = 0.f;
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        const float av = arow[c];
        if (real_row) mx = fmaxf(mx, av);
        if (c < C) {
            const size_t idx = (size_t)((SKEW * me + c) % G) * LANES + (me & 63);
            sw[idx] = srow[c];
            aw[idx] = av;
        } else {
            state_nyq[(size_t)b * TpPad + me] = srow[c];
            amp_nyq[(size_t)b * TpPad + me] = av;
        }
    }
    // block max -> atomic max on the bit pattern (non-negative floats order like unsigned ints)
    __shared__ float red[256];
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0 && real_row) atomicMax(amax_bits + b, __float_as_uint(red[0]));
}

__global__ void __launch_bounds__(256) k_from_skew(float2 *state, const float2 *state_w, const float2 *state_nyq, int T,
                                                    int F, int L, int Q, int G, int TpPad) {
    const int me = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), C = F - 1;
    float2 *orow = state + ((size_t)b * Tp + me) * Np;
    const float2 *sw = state_w + (size_t)b * G * LANES;
    for (int n = threadIdx.x; n < Np; n += blockDim.x) {
        int c = n - L;
        bool conj = false;
        if (c < 0) { c = -c; conj = true; }
        else if (c > C) { c = 2 * C - c; conj = true; }
        float2 v = (c < C) ? sw[(size_t)((SKEW * me + c) % G) * LANES + (me & 63)] : state_nyq[(size_t)b * TpPad + me];
        if (conj) v.y = -v.y;
        orow[n] = v;
    }
}

constexpr uint32_t mask_all(int Q, int L) { return (Q * (L + 1) >= 32) ? 0xffffffffu : ((1u << (Q * (L + 1))) - 1u); }

template <int Q, int L, uint32_t MASK> hipError_t launch_k(const SysArgs &a, int B, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_systolic<Q, L, MASK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_systolic<Q, L, MASK>), dim3(B), dim3(NTHREADS), LDS_BYTES, s, a);
    return hipGetLastError();
}

// mask bit r*(L+1)+k set <=> |W[0][r][k]| > 1e-12.  Default sqrt-Hann windows give these patterns (L = 5):
constexpr uint32_t MASK_Q4_L5_DEFAULT = 0b111111'010111'111111'000011u;  // (r=3 | r=2 | r=1 | r=0), 6 bits each, bit k: r=0:{0,1} r=1:all r=2:{0,1,2,4} r=3:all
constexpr uint32_t MASK_Q2_L5_DEFAULT = 0b010111'000011u;                            // r=0:{0,1} r=1:{0,1,2,4}

struct Tables {
    int Q, L;
    uint32_t mask;
    float w[64];
};

}  // namespace

// =============================================================================================
// host side
// =============================================================================================
hipError_t systolic_build(SystolicPlan &sp, int F, int L, int Q, int Qp, const double *const W[3]) {
    sp.F = F; sp.L = L; sp.Q = Q;
    for (int i = 0; i < 3; ++i) sp.ok[i] = false;
    const int C = F - 1;
    if (Qp != Q || !(Q == 2 || Q == 4) || L != 5) return hipSuccess;
    if (C % 8 != 0 || C > ROWP || C < 16) return hipSuccess;
    if ((Q - 1) * SKEW + L + 1 > LAG) return hipSuccess;
    const int K1 = L + 1;
    for (int i = 0; i < 3; ++i) {
        if (!W[i]) continue;
        // twiddle structure: W[p][r][k] == W[0][r][k] * exp(2j*pi*p*r/Q)
        bool ok = true;
        double scale = 0;
        for (int x = 0; x < Q * Q * K1; ++x) scale = std::fmax(scale, std::hypot(W[i][2 * x], W[i][2 * x + 1]));
        for (int p = 0; p < Q && ok; ++p)
            for (int r = 0; r < Q && ok; ++r)
                for (int k = 0; k <= L; ++k) {
                    if (r == 0 && k == 0) continue;  // never read by the kernels
                    const double ang = 2.0 * M_PI * p * r / Q;
                    const double br = W[i][2 * ((0 * Q + r) * K1 + k)], bi = W[i][2 * ((0 * Q + r) * K1 + k) + 1];
                    const double er = br * std::cos(ang) - bi * std::sin(ang), ei = br * std::sin(ang) + bi * std::cos(ang);
                    const double wr = W[i][2 * ((p * Q + r) * K1 + k)], wi = W[i][2 * ((p * Q + r) * K1 + k) + 1];
                    if (std::hypot(wr - er, wi - ei) > 1e-9 * scale) { ok = false; break; }
                }
        if (!ok) continue;
        Tables *tb = new Tables();
        tb->Q = Q; tb->L = L; tb->mask = 0;
        for (int r = 0; r < Q; ++r)
            for (int k = 0; k <= L; ++k) {
                const double wr = W[i][2 * (r * K1 + k)], wi = W[i][2 * (r * K1 + k) + 1];
                const bool on = std::hypot(wr, wi) > 1.0e-12;  // lws.pyx:231-232
                if (on) tb->mask |= 1u << (r * K1 + k);
                tb->w[2 * (r * K1 + k)] = on ? (float)wr : 0.f;
                tb->w[2 * (r * K1 + k) + 1] = on ? (float)wi : 0.f;
            }
        sp.tables[i] = tb;
        sp.ok[i] = true;
    }
    return hipSuccess;
}

void systolic_release(SystolicPlan &sp) {
    for (int i = 0; i < 3; ++i) {
        delete static_cast<Tables *>(sp.tables[i]);
        sp.tables[i] = nullptr;
        sp.ok[i] = false;
    }
    if (sp.sk_state) (void)hipFree(sp.sk_state);
    if (sp.sk_amp) (void)hipFree(sp.sk_amp);
    sp.sk_state = sp.sk_amp = nullptr;
    sp.sk_state_cap = sp.sk_amp_cap = 0;
}

bool systolic_supports(const SystolicPlan &sp, int wsel, int T) {
    return wsel >= 0 && wsel < 3 && sp.ok[wsel] && T >= 1;  // iteration-count limit: SYSTOLIC_MAX_ITERS
}

const char *systolic_name(const SystolicPlan &sp) { return sp.name; }

hipError_t launch_systolic(SystolicPlan &sp, int wsel, float2 *state, const float *amp, const float *thr, int B,
                           int T, int iters, hipStream_t stream, int *launches, hipEvent_t ev0, hipEvent_t ev1) {
    const Tables *tb = static_cast<const Tables *>(sp.tables[wsel]);
    const int Q = sp.Q, L = sp.L, F = sp.F;
    const int Tp = T + 2 * (Q - 1);
    const int Kr = (Tp + LANES - 1) / LANES;
    const int G = ROWP * Kr;
    const int TpPad = (Tp + 63) & ~63;
    // scratch: state_w, state_nyq | amp_w, amp_nyq, amax
    const size_t n_w = (size_t)B * G * LANES, n_n = (size_t)B * TpPad;
    const size_t need_s = (n_w + n_n) * sizeof(float2);
    const size_t need_a = (n_w + n_n) * sizeof(float) + (size_t)B * sizeof(unsigned);
    hipError_t e;
    if (need_s > sp.sk_state_cap) {
        if (sp.sk_state) (void)hipFree(sp.sk_state);
        sp.sk_state = nullptr; sp.sk_state_cap = 0;
        if ((e = hipMalloc(&sp.sk_state, need_s)) != hipSuccess) return e;
        sp.sk_state_cap = need_s;
    }
    if (need_a > sp.sk_amp_cap) {
        if (sp.sk_amp) (void)hipFree(sp.sk_amp);
        sp.sk_amp = nullptr; sp.sk_amp_cap = 0;
        if ((e = hipMalloc(&sp.sk_amp, need_a)) != hipSuccess) return e;
        sp.sk_amp_cap = need_a;
    }
    float2 *state_w = static_cast<float2 *>(sp.sk_state);
    float2 *state_nyq = state_w + n_w;
    float *amp_w = static_cast<float *>(sp.sk_amp);
    float *amp_nyq = amp_w + n_w;
    unsigned *amax_bits = reinterpret_cast<unsigned *>(amp_nyq + n_n);
    if ((e = hipMemsetAsync(amax_bits, 0, (size_t)B * sizeof(unsigned), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_to_skew, dim3(Tp, B), dim3(256), 0, stream, state, amp, state_w, amp_w, state_nyq, amp_nyq,
                       amax_bits, T, F, L, Q, G, TpPad);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    int nl = 0;
    if (ev0) (void)hipEventRecord(ev0, stream);
    if (iters > MAX_ITERS) return hipErrorInvalidValue;  // caller checks SYSTOLIC_MAX_ITERS
    for (int i0 = 0; i0 < iters; i0 += MAX_ITERS) {
        SysArgs a;
        a.state_w = state_w; a.amp_w = amp_w; a.state_nyq = state_nyq; a.amp_nyq = amp_nyq;
        a.thr = thr + i0; a.amax = reinterpret_cast<const float *>(amax_bits);
        a.n_iters = (iters - i0 < MAX_ITERS) ? iters - i0 : MAX_ITERS;
        a.T = T; a.Tp = Tp; a.TpPad = TpPad; a.Kr = Kr; a.G = G; a.C = F - 1;
        for (int x = 0; x < 64; ++x) a.w[x] = x < 2 * Q * (L + 1) ? tb->w[x] : 0.f;
        if (Q == 4) {
            if (tb->mask == MASK_Q4_L5_DEFAULT) { e = launch_k<4, 5, MASK_Q4_L5_DEFAULT>(a, B, stream); sp.name = "systolic_q4_l5_hannmask"; }
            else { e = launch_k<4, 5, mask_all(4, 5)>(a, B, stream); sp.name = "systolic_q4_l5_allmask"; }
        } else {
            if (tb->mask == MASK_Q2_L5_DEFAULT) { e = launch_k<2, 5, MASK_Q2_L5_DEFAULT>(a, B, stream); sp.name = "systolic_q2_l5_hannmask"; }
            else { e = launch_k<2, 5, mask_all(2, 5)>(a, B, stream); sp.name = "systolic_q2_l5_allmask"; }
        }
        if (e != hipSuccess) return e;
        ++nl;
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    hipLaunchKernelGGL(k_from_skew, dim3(Tp, B), dim3(256), 0, stream, state, state_w, state_nyq, T, F, L, Q, G, TpPad);
    if (launches) *launches = nl;
    return hipGetLastError();
}

}  // namespace lws
