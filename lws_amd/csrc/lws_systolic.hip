// lws_systolic.hip -- the fast path for batch LWS (LWSQ2 / LWSQ4 / LWSanyQ, lwslib.cpp:72-373) on
// gfx950: an order-exact *systolic* re-statement of the in-place Gauss-Seidel sweep.
//
// One workgroup (8 waves = 2 per SIMD; the last one also carries the service duties: HBM loader and the Nyquist
// bins) owns one spectrogram and keeps I = 8 consecutive sweeps in flight.  A lane is a (sweep, frame) processor that marches along the bins of its frame,
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
// the weights and the ring slot -- is static).  HBM is touched once per I sweeps: the service wave
// streams the spectrogram in (set 0) ahead of wave 0, the last wave streams it out, both through a
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
// ring entry of production time nu, lane l:  set + ((nu >> 1) & 15) * PAIR_BYTES + l * 16 + (nu & 1) * 8
// -- two consecutive times of one lane share a 16-byte cell, so a reader fetches two adjacent taps with one
// ds_read_b128 (which, unlike ds_read_b64, reaches the LDS peak rate at 1-2 waves per SIMD)
constexpr int SLOT_BYTES = LANES * 8;                    // bytes per production time (64 float2)
constexpr int PAIR_BYTES = 2 * SLOT_BYTES;               // two consecutive times x 64 lanes
constexpr int BLK_BYTES = 8 * SLOT_BYTES;                // one block of 8 steps
constexpr int LANE_B = 16;
constexpr int SET_BYTES = RING * SLOT_BYTES;             // 16 KiB
#ifndef LWS_NSLOTS
#define LWS_NSLOTS 7
#endif
#ifndef LWS_PF
#define LWS_PF 8
#endif
constexpr int NSLOTS = LWS_NSLOTS;                       // sweeps in flight (compute waves)
constexpr int NSETS = NSLOTS + 1;
constexpr int NYQ_OFF = NSETS * SET_BYTES;               // Nyquist values: [set][lane] float2
constexpr int THR_OFF = NYQ_OFF + NSETS * SLOT_BYTES;    // effective thresholds: floats
constexpr int MAX_ITERS = 440;
constexpr int META_OFF = THR_OFF + MAX_ITERS * 4;        // n_eff
constexpr int DONE_OFF = META_OFF + 16;               // per-wave count of completed steps (flow control)
constexpr int LDS_BYTES = DONE_OFF + 64;
constexpr int SKEW = 8, ROWP = SKEW * LANES, LAG = 32, PF = LWS_PF;  // PF: global prefetch distance (4 or 8 steps)
#ifndef LWS_SERVICE_WAVE
#define LWS_SERVICE_WAVE 1   // 1: loader + Nyquist bins on a wave of their own; 0: carried by the last compute wave
#endif
constexpr int NTHREADS = LANES * (NSLOTS + LWS_SERVICE_WAVE);
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
enum { K_RING = 0, K_NYQ = 1, K_SELF = 2, K_NEXT = 3 };  // K_SELF / K_NEXT: previous-sweep value of the own bin c / c+1 (prefetched registers)
struct Src { int kind, set_new, off, conj; };

__host__ __device__ constexpr Src src_normal(int dr, int dk) {
    if (dr == 0 && dk == 1) return Src{K_NEXT, 0, 0, 0};
    const bool is_new = dr < 0 || (dr == 0 && dk < 0);
    return Src{K_RING, is_new ? 1 : 0, SKEW * dr + dk - (is_new ? 0 : LAG), 0};
}
// reader at bin c = P (first bins of a frame), tap at c + dk < 0: image of bin -(c+dk), conjugated
__host__ __device__ constexpr Src src_start(int P, int dr, int dk) {
    const int cm = -(P + dk);            // mirrored bin, 1..L
    const int off0 = SKEW * dr + cm - P;
    if (dr == 0 && cm == P) return Src{K_SELF, 0, 0, 1};
    if (dr == 0 && cm == P + 1) return Src{K_NEXT, 0, 0, 1};
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
    if (dr == 0 && d == 1) return Src{K_NEXT, 0, 0, 1};
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

// volatile: keeps every tap a separate ds_read_b64 (the backend otherwise fuses pairs into
// ds_read2st64_b64, which moves half the bytes per LDS cycle -- MI355X_MICROARCH.md, LDS table)
#ifndef LWS_DBG_NOLDS
#define LWS_DBG_NOLDS 0     // timing experiment: taps come from registers instead of LDS (results invalid)
#endif
#ifndef LWS_DBG_NOMATH
#define LWS_DBG_NOMATH 0    // timing experiment: one add per tap pair instead of the weighted sum (results invalid)
#endif
__device__ __forceinline__ float2 lds_read(int addr) {
#if LWS_DBG_NOLDS
    return make_float2(__int_as_float(addr), 1.0f);
#endif
    // `addr` is a byte offset into the dynamic LDS segment, which starts at LDS address 0 (the kernel
    // has no static __shared__ objects); address space 3 keeps it a ds_ instruction
    using lds_u64 = const volatile __attribute__((address_space(3))) unsigned long long;
    const unsigned long long u = *(lds_u64 *)(unsigned)addr;
    return make_float2(__uint_as_float((unsigned)(u & 0xffffffffull)), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ void lds_write(int addr, float2 v) {
    // volatile, like the reads: program order of all ring traffic is what the flow control below relies on
    using lds_u64w = volatile __attribute__((address_space(3))) unsigned long long;
    *(lds_u64w *)(unsigned)addr = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
}
__device__ __forceinline__ int lds_read_i32(int addr) {
    using lds_i32 = const volatile __attribute__((address_space(3))) int;
    return *(lds_i32 *)(unsigned)addr;
}
__device__ __forceinline__ void lds_write_i32(int addr, int v) {
    using lds_i32w = volatile __attribute__((address_space(3))) int;
    *(lds_i32w *)(unsigned)addr = v;
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f lds_read128(int addr) {
#if LWS_DBG_NOLDS
    return (v4f){__int_as_float(addr), 1.0f, 2.0f, __int_as_float(addr + 1)};
#endif
    using lds_v4 = const volatile __attribute__((address_space(3))) v4f;
    return *(lds_v4 *)(unsigned)addr;
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

// Workgroup barrier between steps.  Every cross-wave read is at least 3 steps younger than its write except the
// Nyquist lane's taps of bins C-1, C-2 (written in phases 7 and 6, read in phase 0), and no entry is read later
// than 30 steps after it was produced while its slot is rewritten after 32: a barrier after every odd phase
// therefore separates every cross-wave write->read and read->overwrite pair (LWS_BARRIER_EVERY=1: every step).
#ifndef LWS_BARRIER_EVERY
#define LWS_BARRIER_EVERY 2
#endif
template <int P> __device__ __forceinline__ void step_barrier() {
    if constexpr (LWS_BARRIER_EVERY == 1 || (P & 1)) __syncthreads();
}

// Barrier-free alternative (LWS_FLOW=1, default): every wave publishes how many steps it has completed; before a
// step a wave only waits for the waves it actually exchanges data with --
//   data:       its producer (previous slot, or the service wave's loader / Nyquist lanes) must have completed step s-3
//   overwrite:  its consumer (next slot) must have completed step s-2 (ring entries are read for at most 30 steps)
// so waves drift by a step or two against each other and the two waves of a SIMD stop bursting LDS reads and
// arithmetic at the same moments.  LDS executes the operations of a wave in program order, all ring accesses are
// volatile (compiler order), so "write data, then the counter" / "read the counter, then the data" is sufficient.
#ifndef LWS_FLOW
#define LWS_FLOW 1
#endif
__device__ __forceinline__ void flow_wait(int lane, int s, int req_off) {
    // lane l < NWAVES watches wave l; req_off: how many steps behind s that wave may be (large = don't care)
    const int addr = DONE_OFF + (lane & 15) * 4;
    while (true) {
        const int v = lds_read_i32(addr);
        if (__all(v >= s - req_off)) break;
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ __forceinline__ void flow_publish(int lane, int wave, int s_done) {
    if (lane == 0) lds_write_i32(DONE_OFF + wave * 4, s_done);
}

// Per-lane registers of a compute lane that stay valid for one block of 8 steps.
struct LaneCtx {
    int nb[4][4];     // [d][m]: LDS address of lane (rho - d) in the own (new) set, block (a - m) & 3
    int ob[4][4];     // [d][m]: lane (rho + d) in the previous sweep's (old) set
    int nyq_base;     // NYQ_OFF + own set row + lane*8 (taps derive the neighbour lane / set from it)
    int lane8;
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
    return base[m] + (within >> 1) * PAIR_BYTES + (within & 1) * 8;
}

template <int P, int DR, int DK, int EDGE>  // EDGE: 0 normal, 1 frame start, 2 frame end
__device__ __forceinline__ float2 tap(const LaneCtx &cx, float2 self_old, float2 next_old) {
    constexpr Src s = (EDGE == 0) ? src_normal(DR, DK) : (EDGE == 1 ? src_start(P, DR, DK) : src_end(P, DR, DK));
    constexpr int d = DR < 0 ? -DR : DR;
    float2 v;
    if constexpr (s.kind == K_SELF) v = self_old;
    else if constexpr (s.kind == K_NEXT) v = next_old;
    else if constexpr (s.kind == K_NYQ) {
        // rare (one lane, last bins of a frame): Nyquist value of frame m+DR in the own / previous set
        const int ln = (cx.lane8 + 8 * DR) & (SLOT_BYTES - 1);
        v = lds_read(cx.nyq_base - cx.lane8 + ln - (s.set_new ? 0 : SLOT_BYTES));
    }
    else {
        static_assert(s.off <= -1 && s.off >= -30, "tap outside ring retention (ages 1..30)");
        v = lds_read(s.set_new ? ring_addr<P, s.off>(cx.nb[d]) : ring_addr<P, s.off>(cx.ob[d]));
    }
    if constexpr (s.conj) v = cj(v);
    return v;
}

// Normal-source taps of frame m+DR (DR != 0), bins c-L .. c+L: they are consecutive in production time, so
// two of them come with each ds_read_b128.  KMASK bit |dk| says whether the tap is used.
template <int P, int DR, int L, uint32_t KMASK>
__device__ __forceinline__ void load_row(const LaneCtx &cx, float2 (&t)[2 * L + 1]) {
    constexpr int d = DR < 0 ? -DR : DR;
    constexpr int base_off = SKEW * DR - (DR > 0 ? LAG : 0);
    constexpr int q_lo = P + base_off - L;                       // first tap, relative production time
    constexpr int q_first = (q_lo >= 0) ? (q_lo & ~1) : -(((-q_lo) + 1) & ~1);   // round down to even
    static_for<L + 2>([&](auto ip) {
        constexpr int q = q_first + 2 * decltype(ip)::value;     // even
        constexpr int dka = q - P - base_off, dkb = dka + 1;
        constexpr bool in_a = dka >= -L && dka <= L, in_b = dkb >= -L && dkb <= L;
        constexpr bool need_a = in_a && ((KMASK >> (dka < 0 ? -dka : dka)) & 1u);
        constexpr bool need_b = in_b && ((KMASK >> (dkb < 0 ? -dkb : dkb)) & 1u);
        if constexpr (need_a || need_b) {
            static_assert(q >= -32 && q + 1 <= 7, "ring retention exceeded");
            constexpr int fl = (q >= 0) ? 0 : -((-q + 7) / 8);
            constexpr int m = (-fl) & 3;
            constexpr int within = q - 8 * fl;                   // even, 0..6
            const int addr = (DR < 0 ? cx.nb[d][m] : cx.ob[d][m]) + (within >> 1) * PAIR_BYTES;
            if constexpr (need_a && need_b) {
                const v4f v = lds_read128(addr);
                t[dka + L] = make_float2(v.x, v.y);
                t[dkb + L] = make_float2(v.z, v.w);
            } else if constexpr (need_a) {
                t[dka + L] = lds_read(addr);
            } else {
                t[dkb + L] = lds_read(addr + 8);
            }
        }
    });
}

// acc += w*b + conj(w)*c with w = (wr, wi) * j^ROT   (grouped form of lwslib.cpp:310-311)
#ifndef LWS_PKMATH
#define LWS_PKMATH 0
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
template <int ROT> __device__ __forceinline__ void pair_rot(float2 &a, float wr, float wi, float2 b, float2 c) {
#if LWS_DBG_NOMATH
    a.x += b.x; a.y += c.y; return;
#endif
#if LWS_PKMATH
    // packed fp32: (sx, sy) = b + c, (dx, dy) = b - c, then two v_pk_fma_f32
    const v2f vb = {b.x, b.y}, vc = {c.x, c.y};
    const v2f sm = vb + vc, df = vb - vc;
    const v2f dsw = {df.y, df.x};
    v2f acc = {a.x, a.y};
    v2f w1, w2;
    if constexpr (ROT == 0) { w1 = (v2f){wr, wr}; w2 = (v2f){-wi, wi}; }
    else if constexpr (ROT == 1) { w1 = (v2f){-wi, -wi}; w2 = (v2f){-wr, wr}; }
    else if constexpr (ROT == 2) { w1 = (v2f){-wr, -wr}; w2 = (v2f){wi, -wi}; }
    else { w1 = (v2f){wi, wi}; w2 = (v2f){wr, -wr}; }
    acc = __builtin_elementwise_fma(w1, sm, acc);
    acc = __builtin_elementwise_fma(w2, dsw, acc);
    a.x = acc.x; a.y = acc.y;
#else
    const float sx = b.x + c.x, dy = b.y - c.y, sy = b.y + c.y, dx = b.x - c.x;
    // two fused multiply-adds per component (weights are wave-uniform scalars)
    if constexpr (ROT == 0) { a.x = fmaf(-wi, dy, fmaf(wr, sx, a.x)); a.y = fmaf(wi, dx, fmaf(wr, sy, a.y)); }
    else if constexpr (ROT == 1) { a.x = fmaf(-wr, dy, fmaf(-wi, sx, a.x)); a.y = fmaf(wr, dx, fmaf(-wi, sy, a.y)); }
    else if constexpr (ROT == 2) { a.x = fmaf(wi, dy, fmaf(-wr, sx, a.x)); a.y = fmaf(-wi, dx, fmaf(-wr, sy, a.y)); }
    else { a.x = fmaf(wr, dy, fmaf(wi, sx, a.x)); a.y = fmaf(-wr, dx, fmaf(wi, sy, a.y)); }
#endif
}
__device__ __forceinline__ float2 cadd(float2 p, float2 q) {
#if LWS_PKMATH
    const v2f r = (v2f){p.x, p.y} + (v2f){q.x, q.y};
    return make_float2(r.x, r.y);
#else
    return make_float2(p.x + q.x, p.y + q.y);
#endif
}
__device__ __forceinline__ float2 csub(float2 p, float2 q) {
#if LWS_PKMATH
    const v2f r = (v2f){p.x, p.y} - (v2f){q.x, q.y};
    return make_float2(r.x, r.y);
#else
    return make_float2(p.x - q.x, p.y - q.y);
#endif
}

// The weighted sum of one bin at phase P (bin % 8 == P), all taps with compile-time ring offsets.
template <int Q, int L, uint32_t MASK, int P>
__device__ __forceinline__ float2 weighted_sum(const SysArgs &a, const LaneCtx &cx, float2 self_old, float2 next_old) {
    constexpr int K1 = L + 1;
    float2 acc = make_float2(0.f, 0.f);
    // centre frame: W[.,0,k] does not depend on bin % Q
    {
        float2 lo[L], hi[L];
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> k) & 1u) {
                lo[k - 1] = tap<P, 0, -k, 0>(cx, self_old, next_old);
                hi[k - 1] = tap<P, 0, k, 0>(cx, self_old, next_old);
            }
        });
        if constexpr (P < L) {
            if (cx.is_start)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> k) & 1u) && (P - k < 0)) lo[k - 1] = tap<P, 0, -k, 1>(cx, self_old, next_old);
                });
        }
        if constexpr (P + L >= 8) {
            if (cx.is_end)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> k) & 1u) && (P + k >= 8)) hi[k - 1] = tap<P, 0, k, 2>(cx, self_old, next_old);
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
        constexpr uint32_t kmask = (MASK >> (r * K1)) & ((1u << K1) - 1u);
        load_row<P, -r, L, kmask>(cx, up);
        load_row<P, r, L, kmask>(cx, dn);
        if constexpr (P < L) {
            if (cx.is_start)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> (r * K1 + k)) & 1u) && (P - k < 0)) {
                        up[L - k] = tap<P, -r, -k, 1>(cx, self_old, next_old);
                        dn[L - k] = tap<P, r, -k, 1>(cx, self_old, next_old);
                    }
                });
        }
        if constexpr (P + L >= 8) {
            if (cx.is_end)
                static_for<L>([&](auto ik) {
                    constexpr int k = decltype(ik)::value + 1;
                    if constexpr (((MASK >> (r * K1 + k)) & 1u) && (P + k >= 8)) {
                        up[L + k] = tap<P, -r, k, 2>(cx, self_old, next_old);
                        dn[L + k] = tap<P, r, k, 2>(cx, self_old, next_old);
                    }
                });
        }
        float2 accr = make_float2(0.f, 0.f);  // one accumulator per frame pair: independent dependency chains
        if constexpr ((MASK >> (r * K1)) & 1u)
            pair_rot<rot>(accr, a.w[2 * (r * K1)], a.w[2 * (r * K1) + 1], up[L], dn[L]);
        static_for<L>([&](auto ik) {
            constexpr int k = decltype(ik)::value + 1;
            if constexpr ((MASK >> (r * K1 + k)) & 1u) {
                // W[mod]*S[m-r,c-k] + conj(W[mod])*S[m+r,c-k] + W[-mod]*S[m+r,c+k] + conj(W[-mod])*S[m-r,c+k]
                // with W[-mod] = +-W[mod] for a real / imaginary twiddle (the LWSQ2 / LWSQ4 grouping)
                const float wr = a.w[2 * (r * K1 + k)], wi = a.w[2 * (r * K1 + k) + 1];
                float2 b, c;
                if constexpr ((rot & 1) == 0) { b = cadd(up[L - k], dn[L + k]); c = cadd(dn[L - k], up[L + k]); }
                else { b = csub(up[L - k], dn[L + k]); c = csub(dn[L - k], up[L + k]); }
                pair_rot<rot>(accr, wr, wi, b, c);
            }
        });
        acc = cadd(acc, accr);
    });
    return acc;
}

// Magnitude re-projection (lwslib.cpp:356-360): keep the old value unless the bin is active and the sum is
// non-zero.  target/|acc| is evaluated as target * rsqrt(|acc|^2) with one Newton step on the hardware
// reciprocal square root (relative error < 2^-22, the size of the two roundings of sqrt-then-divide);
// sums too small to square in fp32 are rescaled first, so "|acc| > 0" keeps the reference's meaning.
__device__ __forceinline__ float2 project(float2 acc, float target, bool active, float2 old) {
    float m2 = acc.x * acc.x + acc.y * acc.y;
    const bool tiny = m2 < 1e-30f;
    const float ax = tiny ? acc.x * 0x1p60f : acc.x, ay = tiny ? acc.y * 0x1p60f : acc.y;
    m2 = tiny ? ax * ax + ay * ay : m2;
    const bool ok = active && (m2 > 0.f);
    float r = __frsqrt_rn(m2);
    r = r * fmaf(-0.5f * m2 * r, r, 1.5f);
    const float sc = target * r;
    return ok ? make_float2(ax * sc, ay * sc) : old;
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
__device__ __forceinline__ void compute_step(const SysArgs &a, const LaneCtx &cx, int lane, int vmod /* clock mod G at phase 0 */,
                                             float2 &self_old, float2 &next_old, const float (&amp_cur)[8], float2 *state_w_b) {
    // taps, weighted sum, projection
    const float2 acc = weighted_sum<Q, L, MASK, P>(a, cx, self_old, next_old);
    const float target = amp_cur[P];
    const bool active = cx.live && (target > cx.thr);
    const float2 out = project(acc, target, active, self_old);
    // publish: own set, slot (v mod 32) = block m = 0, within = P
    lds_write(cx.nb[0][0] + (P >> 1) * PAIR_BYTES + (P & 1) * 8, out);
    if (cx.store) state_w_b[(size_t)(vmod + P) * LANES + lane] = out;   // G is a multiple of 8: no wrap inside a block
    // prefetch: previous-sweep value of the own bin two steps ahead (age 30 now: no ring read is ever older,
    // which leaves two steps between the last read of an entry and its overwrite)
    self_old = next_old;
    next_old = lds_read(ring_addr<P, -30>(cx.ob[0]));
}

// State of the service duties (HBM loader for set 0 and the Nyquist bins of every sweep slot).
struct ServiceState {
    float2 pend[PF];        // loader: values in flight from HBM
    float nyq_amp_next;     // Nyquist lanes: target magnitude of the next block's Nyquist bin
    float2 nyq_in_next;     // Nyquist loader lane: previous-sweep Nyquist value of the next block's frame
};

// Phase 0 of a block: lane l < NSLOTS computes the Nyquist bin (bin C = F-1) of the frame of sweep slot l whose
// 512-step period has just ended; lane NSLOTS feeds set 0 with the stored Nyquist value of that frame.
template <int Q, int L, uint32_t MASK>
__device__ __forceinline__ void service_nyquist(const SysArgs &a, ServiceState &sv, int lane, int t0, int n_eff,
                                                int n_groups, const float *thr_eff, float2 *state_nyq_b,
                                                const float *amp_nyq_b) {
    constexpr int K1 = L + 1;
    const int C = a.C, Kr = a.Kr;
    const int ablk = (t0 >> 3);
    const int slot = lane;                       // lanes 0..NSLOTS-1
    const bool is_nyq_lane = lane < NSLOTS;
    const bool is_nyq_loader = lane == NSLOTS;
    const int v0 = t0 - (is_nyq_lane ? (slot + 1) * LAG : 0);
    const int vrow = (v0 - C) >> 3;              // virtual frame whose Nyquist bin is due now
    const int rho = vrow & 63, kap = vrow >> 6;
    const int g = kap / Kr, k = kap - g * Kr;
    const int me = k * LANES + rho;
    const int j = g * NSLOTS + (is_nyq_lane ? slot : -1);
    const bool valid = (v0 - C >= 0) && (me < a.Tp) && (is_nyq_lane ? (j < n_eff) : (is_nyq_loader && g < n_groups));
    if (is_nyq_loader) {
        lds_write(NYQ_OFF + rho * 8, sv.nyq_in_next);  // loaded one block ago for this frame
        const int vr1 = vrow + 1, rho1 = vr1 & 63, kap1 = vr1 >> 6;
        const int g1 = kap1 / Kr, k1 = kap1 - g1 * Kr, me1 = k1 * LANES + rho1;
        if (vr1 >= 0 && me1 < a.Tp) sv.nyq_in_next = load_l2(state_nyq_b + me1);
    }
    if (is_nyq_lane) {
        const float target = sv.nyq_amp_next;
        const bool real_row = valid && (me >= Q - 1) && (me < a.T + Q - 1);
        const float thr = thr_eff[valid ? j : 0];
        const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
        int nb[4][4], ob[4][4], nn[4], no[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int ln = ((rho - d) & 63), lo = ((rho + d) & 63);
            nn[d] = NYQ_OFF + (slot + 1) * SLOT_BYTES + ln * 8;
            no[d] = NYQ_OFF + slot * SLOT_BYTES + lo * 8;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int blk = ((ablk - m) & 3) * BLK_BYTES;
                nb[d][m] = set_new + blk + ln * LANE_B;
                ob[d][m] = set_old + blk + lo * LANE_B;
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
        sv.nyq_amp_next = (vr1 >= 0 && me1 < a.Tp) ? amp_nyq_b[me1] : 0.f;
    }
}

template <int Q, int L, uint32_t MASK>
__global__ void __launch_bounds__(NTHREADS, (NTHREADS + 255) / 256) k_systolic(SysArgs a) {
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
    if (threadIdx.x < 16) reinterpret_cast<int *>(smem + DONE_OFF)[threadIdx.x] = 0;
    __syncthreads();
    const int n_eff = meta[0];
    if (n_eff == 0) return;
    const int n_groups = (n_eff + NSLOTS - 1) / NSLOTS;
    // slot i runs on clock v_i = t - (i+1)*LAG; the loader (virtual slot -1) on clock t
    const int t_end = (n_groups - 1) * G + (NSLOTS + 1) * LAG + SKEW * a.Tp + ROWP + 8;

    const bool is_compute = wave < NSLOTS;
    const bool is_service = LWS_SERVICE_WAVE ? (wave == NSLOTS) : (wave == NSLOTS - 1);
    const int slot = wave;
    LaneCtx cx;
    float2 self_old = make_float2(0.f, 0.f), next_old = make_float2(0.f, 0.f);
    // target magnitudes of the current block's 8 bins and (in flight) of the next block's: all 8 global loads of a
    // block are issued together one block ahead, so no step ever waits on HBM latency
    float amp_cur[8], amp_nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) amp_cur[i] = amp_nxt[i] = 0.f;
    ServiceState sv;
#pragma unroll
    for (int i = 0; i < PF; ++i) sv.pend[i] = make_float2(0.f, 0.f);
    sv.nyq_amp_next = 0.f;
    sv.nyq_in_next = make_float2(0.f, 0.f);
    if (is_service) {
#pragma unroll
        for (int i = 0; i < PF; ++i) sv.pend[i] = load_l2(state_w_b + (size_t)(i % G) * LANES + lane);  // clocks 0..7
    }
    const int set_new = (slot + 1) * SET_BYTES, set_old = slot * SET_BYTES;
    // flow control: which waves this wave waits for (lane l watches wave l), and how far behind they may be
    constexpr int NWAVES = NSLOTS + LWS_SERVICE_WAVE, SVC = LWS_SERVICE_WAVE ? NSLOTS : NSLOTS - 1, FAR = 1 << 29;
    int req_off = FAR, req_off_p0 = FAR;
    if (lane < NWAVES) {
        if (LWS_SERVICE_WAVE && wave == SVC) {
            // loader overwrites what slot 0 still reads; Nyquist lanes read last step's bins of every slot at phase 0
            req_off = (lane == 0) ? 1 : FAR;
            req_off_p0 = (lane < NSLOTS) ? 0 : FAR;
        } else {
            const int producer = (wave == 0) ? SVC : wave - 1;
            if (lane == producer || lane == SVC) req_off = 2;
            if (lane == wave + 1 && lane < NSLOTS) req_off = 1;
            if (lane == wave) req_off = FAR;
            req_off_p0 = req_off;
        }
    }
    for (int t0 = 0; t0 < t_end; t0 += 8) {
        const int v0 = t0 - (slot + 1) * LAG;  // clock of this sweep slot at phase 0 of the block (multiple of 8)
        // ---- block prologue: where is this lane?
        const int vv = v0 - SKEW * lane;        // clock relative to the start of lane's first frame
        const int cbase = vv & (ROWP - 1);
        const int kap = vv >> 9;
        const int g = kap / Kr, k = kap - g * Kr;
        const int me = k * LANES + lane;
        const int j = g * NSLOTS + slot;
        const bool valid = is_compute && (vv >= 0) && (j < n_eff) && (me < a.Tp);
        const bool real_row = valid && (me >= Q - 1) && (me < a.T + Q - 1);
        cx.live = real_row && (cbase < C);
        cx.store = valid && (cbase < C) && (slot == NSLOTS - 1 || j == n_eff - 1);
        cx.is_start = (cbase == 0);
        cx.is_end = (cbase == C - 8);
        cx.thr = thr_eff[(valid ? j : 0)];
        const int ablk = (v0 >> 3);
        cx.lane8 = lane * 8;
        cx.nyq_base = NYQ_OFF + (slot + 1) * SLOT_BYTES + lane * 8;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int ln = ((lane - d) & 63) * LANE_B, lo = ((lane + d) & 63) * LANE_B;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int blk = ((ablk - m) & 3) * BLK_BYTES;
                cx.nb[d][m] = set_new + blk + ln;
                cx.ob[d][m] = set_old + blk + lo;
            }
        }
        // one block early, so that the prefetched own-old value and magnitudes are warm at the first bin
        const bool any_valid = is_compute && __any((vv >= -8) && (j < n_eff || vv < 0));
        const int vmod = __builtin_amdgcn_readfirstlane(((v0 % G) + G) % G);  // wave-uniform, once per 8 steps
        const int tmod = __builtin_amdgcn_readfirstlane(t0 % G);
        {
            int vnext = vmod + 8;
            vnext -= (vnext >= G) ? G : 0;     // G is a multiple of 8: the next block does not wrap inside
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                amp_cur[i] = amp_nxt[i];
                amp_nxt[i] = amp_w_b[(size_t)(vnext + i) * LANES + lane];
            }
        }
#if LWS_FLOW
        flow_wait(lane, t0, req_off_p0);
#endif
        // ---- Nyquist bins of all slots fall on phase 0 (C is a multiple of 8)
        if (is_service) service_nyquist<Q, L, MASK>(a, sv, lane, t0, n_eff, n_groups, thr_eff, state_nyq_b, amp_nyq_b);
        // ---- 8 steps, phase static
        static_for<8>([&](auto ip) {
            constexpr int P = decltype(ip)::value;
#if LWS_FLOW
            if constexpr (P > 0) flow_wait(lane, t0 + P, req_off);
#endif
            if (any_valid)
                compute_step<Q, L, MASK, P>(a, cx, lane, vmod, self_old, next_old, amp_cur, state_w_b);
            if (is_service) {
                // loader: feed set 0 with the values the virtual previous sweep would produce, PF steps ahead
                lds_write(((t0 >> 3) & 3) * BLK_BYTES + (P >> 1) * PAIR_BYTES + (P & 1) * 8 + lane * LANE_B, sv.pend[P % PF]);
                int ild = tmod + P + PF;
                ild -= (ild >= G) ? G : 0;
#if LWS_DBG_NOLDS && LWS_DBG_NOMATH
                sv.pend[P % PF] = make_float2((float)ild, 0.f);   // skeleton timing: no HBM latency floor either
#else
                sv.pend[P % PF] = load_l2(state_w_b + (size_t)ild * LANES + lane);
#endif
            }
#if LWS_FLOW
            flow_publish(lane, wave, t0 + P + 1);
#else
            step_barrier<P>();
#endif
        });
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
