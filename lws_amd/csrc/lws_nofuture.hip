// lws_nofuture.hip -- LDS-resident engine for the no-future sweeps (NoFuture_LWSQ2 / Q4 / anyQ, lwslib.cpp:473-690), fp32.
//
// A no-future sweep updates frame m from frames m-1 ... m-Q+1 only (lwslib.cpp:640-671): frames are strictly sequential,
// and inside a frame every bin is independent -- except under the shipped NoFuture_LWSQ4, whose flat offset
// (m-r)*Np + 2n +- k (lwslib.cpp:559-594) runs past the end of frame m-r into the next one, for r = 1 into frame m itself
// at columns 2n - Np +- k < n: the upper half of the bins then depends on earlier bins of the same frame.
// Bins [n0, n1) are still independent of each other as long as every such column lies below n0, i.e. n < (n0 + Np - L)/2,
// so the remaining range halves each round: ~10 parallel rounds per frame instead of ~260 serial bins (same rounds as
// lws_generic.hip, whose fp32 result this kernel reproduces bit for bit: same arithmetic, same order, no contraction).
//
// What this kernel changes is where the data lives: one workgroup keeps a ring of the last NR extended frames of its
// spectrogram in LDS, so the ~33 taps of a bin and the ~10 barrier rounds of a frame run at LDS latency instead of L2 /
// HBM latency (a round of the generic engine costs ~3 us: 17 ms for 256 x 500 x 513; here ~0.2 us).  HBM sees every frame
// once on the way in (prefetched one frame ahead, coalesced) and once on the way out, plus its magnitudes once.
#include "lws_common.h"
#include "lws_nofuture.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace lws {
namespace {

// R: float, or double (round 5: the one-lane-per-bin variant only -- an fp64 plan's no-future sweeps in the generic engine's order, bit for bit)
template <typename R> struct NfArgsT {
    typename cx<R>::type *state;       // [B][Tp][Np]
    const R *amp;    // [B][Tp][Np]
    const R *thr;    // [B][n_thr]
    const typename cx<R>::type *w;     // [Q or Q'][Q][L+1], zero where flagged off
    const uint8_t *flag; // [Q][Q][L+1]
    int F, T, L, Q, n_thr, NR, compat;
    int rows;            // weight rows kept in LDS: Q (summarised tensors: row = bin mod Q) or the period P of a general tensor's rows
                         // (Q' = N rows, one per bin -- lws.pyx:164-181 -- that repeat with period P = frame / gcd(frame, hop): row = bin mod P)
};
using NfArgs = NfArgsT<float>;

template <typename C> __device__ __forceinline__ void pair(C &a, const C w, const C b, const C c) {
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}
template <typename C> __device__ __forceinline__ void mac(C &a, const C w, const C s) {   // a += w * s
    a.x += w.x * s.x - w.y * s.y;
    a.y += w.x * s.y + w.y * s.x;
}
template <typename C> __device__ __forceinline__ void macc(C &a, const C w, const C s) {  // a += conj(w) * s
    a.x += w.x * s.x + w.y * s.y;
    a.y += w.x * s.y - w.y * s.x;
}

// Rows written by this workgroup in an earlier sweep are read back past the CU's L1.
__device__ __forceinline__ float2 load_state(const float2 *p) {
    const unsigned long long u =
        __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)(u & 0xffffffffull)), __uint_as_float((unsigned)(u >> 32)));
}

__device__ __forceinline__ double2 load_state(const double2 *p) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_double2(__longlong_as_double((long long)a), __longlong_as_double((long long)b));
}

// One bin of frame `me`: weighted sum over the past frames, re-projection, Hermitian image upkeep -- update_bin /
// update_bin_nfq4 of lws_generic.hip on the LDS ring.  UNI: every weight row has the same participation mask (m_uni).
template <int QT, int LT, bool COMPAT, bool UNI, typename R, typename C2>
__device__ __forceinline__ void nf_update(int n, int me, R th, const C2 *S, const C2 *W, const unsigned long long *Mk,
                                          unsigned long long m_uni, const R *amp_cur, C2 *cur, int Q_, int L_, int F,
                                          int Np, int NR, int RW_) {
    using float2 = C2;                                  // (the body below is written once for both precisions)
    auto make_float2 = [](R x, R y) { C2 v; v.x = x; v.y = y; return v; };
    const int Q = QT ? QT : Q_, L = LT ? LT : L_, K1 = L + 1, RQ = Q * K1, nyq = F + L - 1;
    const int RW = QT ? QT : RW_;                       // weight rows: a compile-time Q implies a summarised tensor
    const int c = n - L;
    const R target = amp_cur[n];
    if (!(target > th)) return;
    const int row = c % RW;
    const float2 *wa = W + row * RQ;
    const int rowneg = (RW - row) % RW;
    const unsigned long long ma = UNI ? m_uni : Mk[row];
    float2 acc = make_float2((R)0, (R)0);
    if constexpr (COMPAT) {                            // update_bin_nfq4 of lws_generic.hip (lwslib.cpp:550-613)
        const int wrap_at = Np - 2 * n;                 // offsets j >= wrap_at run past the end of the frame
#pragma unroll
        for (int r = Q - 1; r > 0; --r) {
            // flat offset (me - r) * Np + 2n + j: column 2n + j of frame me - r, or past its end in the next frame
            const int i0 = ((me - r) & (NR - 1)) * Np + 2 * n, i1 = ((me - r + 1) & (NR - 1)) * Np + 2 * n - Np;
            const int u = r * K1;
            const R sgn = ((c & 1) && (r & 1)) ? (R)-1 : (R)1;
#pragma unroll
            for (int k = 1; k <= L; ++k)
                if ((ma >> (u + k)) & 1ull) {
                    float2 hi = S[k >= wrap_at ? i1 + k : i0 + k];
                    hi.x *= sgn; hi.y *= sgn;
                    pair(acc, wa[u + k], S[-k >= wrap_at ? i1 - k : i0 - k], hi);
                }
            if ((ma >> u) & 1ull) mac(acc, wa[u], S[0 >= wrap_at ? i1 : i0]);
        }
    } else {                                           // accumulate() of lws_generic.hip with centre = false, two_sided = 1
        const unsigned long long mb = UNI ? m_uni : Mk[rowneg];
        const float2 *wb = W + rowneg * RQ;
#pragma unroll
        for (int r = 1; r < Q; ++r) {
            const float2 *lf = S + (size_t)((me - r) & (NR - 1)) * Np + n;
            const int u = r * K1;
            if ((ma >> u) & 1ull) mac(acc, wa[u], lf[0]);
#pragma unroll
            for (int k = 1; k <= L; ++k) {
                if ((ma >> (u + k)) & 1ull) mac(acc, wa[u + k], lf[-k]);
                if ((mb >> (u + k)) & 1ull) macc(acc, wb[u + k], lf[k]);
            }
        }
    }
    const R mag = sqrt(acc.x * acc.x + acc.y * acc.y);
    if (!(mag > (R)0)) return;
    const float2 v = make_float2(acc.x * target / mag, acc.y * target / mag);
    cur[n] = v;
    const float2 vc = make_float2(v.x, -v.y);            // Hermitian images in the pad columns (lwslib.cpp:362-367)
    if (n >= L + 1 && n < 2 * L + 1) cur[2 * L - n] = vc;
    else if (n >= F - 1 && n < nyq) cur[2 * nyq - n] = vc;
}

// sum over the 8 adjacent lanes of a group (groups are aligned), in data-parallel-primitive moves
__device__ __forceinline__ float group_sum8(float v) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>());    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>());    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>());   // row_half_mirror: the other quad of the 8
    return v;
}

// The same bin by EIGHT lanes (production variant): lane j sums the tap groups j, j+8, ... (group t = (r-1)(L+1) + k: the two
// taps +-k of frame me-r, or its tap 0), the partial sums are combined across the lanes, lane 0 re-projects and writes.
// A round of the chained NoFuture_LWSQ4 rounds is latency: ~33 dependent taps per lane become ~5.  Same arithmetic per tap as
// nf_update; the order of the sum differs (rounding level).
template <int QT, int LT, bool COMPAT, bool UNI>
__device__ __forceinline__ void nf_update_split(int n, int j, int me, float th, const float2 *S, const float2 *W,
                                                const unsigned long long *Mk, unsigned long long m_uni, const float *amp_cur,
                                                float2 *cur, int Q_, int L_, int F, int Np, int NR, int RW_) {
    const int Q = QT ? QT : Q_, L = LT ? LT : L_, K1 = L + 1, RQ = Q * K1, nyq = F + L - 1;
    const int RW = QT ? QT : RW_;
    const int c = n - L;
    const float target = amp_cur[n];
    if (!(target > th)) return;                         // (the same for the 8 lanes of a bin)
    const int row = c % RW;
    const float2 *wa = W + row * RQ;
    const int rowneg = (RW - row) % RW;
    const unsigned long long ma = UNI ? m_uni : Mk[row], mb = UNI ? m_uni : Mk[rowneg];
    const float2 *wb = W + rowneg * RQ;
    const int nterms = (Q - 1) * K1, wrap_at = Np - 2 * n;
    float2 acc = make_float2(0.f, 0.f);
    // tap groups t = j, j + 8, ... < nterms = (Q-1)(L+1); with a run-time L or Q only the bound of shape_of is known
    // (Q (L+1) <= 64: one participation mask per weight row), so eight rounds cover every admitted shape
    constexpr int NIT = (QT && LT) ? ((QT - 1) * (LT + 1) + 7) / 8 : 8;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int t = j + 8 * i;
        if (t < nterms) {
            const int r = t / K1 + 1, k = t - (r - 1) * K1, u = r * K1;
            if constexpr (COMPAT) {
                const int i0 = ((me - r) & (NR - 1)) * Np + 2 * n, i1 = ((me - r + 1) & (NR - 1)) * Np + 2 * n - Np;
                const float sgn = ((c & 1) && (r & 1)) ? -1.f : 1.f;
                if (k == 0) {
                    if ((ma >> u) & 1ull) mac(acc, wa[u], S[0 >= wrap_at ? i1 : i0]);
                } else if ((ma >> (u + k)) & 1ull) {
                    float2 hi = S[k >= wrap_at ? i1 + k : i0 + k];
                    hi.x *= sgn; hi.y *= sgn;
                    pair(acc, wa[u + k], S[-k >= wrap_at ? i1 - k : i0 - k], hi);
                }
            } else {
                const float2 *lf = S + (size_t)((me - r) & (NR - 1)) * Np + n;
                if (k == 0) {
                    if ((ma >> u) & 1ull) mac(acc, wa[u], lf[0]);
                } else {
                    if ((ma >> (u + k)) & 1ull) mac(acc, wa[u + k], lf[-k]);
                    if ((mb >> (u + k)) & 1ull) macc(acc, wb[u + k], lf[k]);
                }
            }
        }
    }
    acc.x = group_sum8(acc.x);
    acc.y = group_sum8(acc.y);
    if (j != 0) return;
    // target / |acc| as target * rsqrt(|acc|^2) (hardware reciprocal square root, 1 ulp: a round is a dependent chain, and the
    // exact square root and the two divisions of the reference form are a third of it); sums too small to square in fp32
    // are rescaled first so that "|acc| > 0" keeps its meaning
    float m2 = acc.x * acc.x + acc.y * acc.y;
    if (m2 < 1e-30f) {
        acc.x *= 0x1p60f; acc.y *= 0x1p60f;
        m2 = acc.x * acc.x + acc.y * acc.y;
    }
    if (!(m2 > 0.f)) return;
    const float sc = target * __frsqrt_rn(m2);
    const float2 v = make_float2(acc.x * sc, acc.y * sc);
    cur[n] = v;
    const float2 vc = make_float2(v.x, -v.y);
    if (n >= L + 1 && n < 2 * L + 1) cur[2 * L - n] = vc;
    else if (n >= F - 1 && n < nyq) cur[2 * nyq - n] = vc;
}

// QT / LT: compile-time Q and L (0: use the run-time values); COMPAT: NoFuture_LWSQ4's flat addressing (Q = 4 only);
// SPLIT: eight lanes per bin (production) or one (the verification variant, bit-identical to the generic engine)
template <int QT, int LT, bool COMPAT, bool SPLIT, typename R = float>
__global__ void __launch_bounds__(SPLIT ? 1024 : 512) k_nofuture(NfArgsT<R> a) {
    static_assert(!SPLIT || std::is_same<R, float>::value, "the eight-lanes-per-bin variant is fp32");
    using float2 = typename cx<R>::type;          // (the body below is written once for both precisions)
    auto make_float2 = [](R x, R y) { typename cx<R>::type v; v.x = x; v.y = y; return v; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Q = QT ? QT : a.Q, L = LT ? LT : a.L;
    const int RW = QT ? QT : a.rows;
    const int F = a.F, T = a.T, NR = a.NR, K1 = L + 1, RQ = Q * K1;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    float2 *S = reinterpret_cast<float2 *>(smem);                 // [NR][Np] ring of extended frames
    float2 *W = S + (size_t)NR * Np;                              // [RW][Q][K1]
    unsigned long long *Mk = reinterpret_cast<unsigned long long *>(W + RW * RQ);   // [RW] + 1: which weights of a row take part
    R *A = reinterpret_cast<R *>(Mk + RW + 1);            // [2][Np] target magnitudes of the current and the next frame
    float2 *gS = a.state + (size_t)b * Tp * Np;
    const R *gA = a.amp + (size_t)b * Tp * Np;
    for (int i = tid; i < RW * RQ; i += nthr) W[i] = a.w[i];
    // The reference tests a flag per weight (lwslib.cpp:302,321,...).  Here the flags of a row are one bit mask (bit r*K1+k),
    // and when every row has the same mask -- always the case for create_weights' tensors, whose rows differ by unit-modulus
    // twiddles -- it is wave-uniform: the tests become scalar branches instead of 33 dependent byte loads per bin.
    for (int rw = tid; rw <= RW; rw += nthr) {
        unsigned long long mk = 0;
        if (rw < RW) {
            for (int x = 0; x < RQ; ++x) mk |= (unsigned long long)(a.flag[rw * RQ + x] != 0) << x;
        } else {
            bool same = true;
            for (int r2 = 1; r2 < RW; ++r2)
                for (int x = 0; x < RQ; ++x) same = same && ((a.flag[r2 * RQ + x] != 0) == (a.flag[x] != 0));
            mk = same ? 1ull : 0ull;
        }
        Mk[rw] = mk;
    }
    __syncthreads();
    const bool uni = Mk[RW] != 0;
    const unsigned long long m_uni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(Mk[0] >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(Mk[0] & 0xffffffffull));
    // bins whose farthest read (m-1)*Np + 2n + L stays inside frame m-1 are independent of frame m (compat rounds)
    int n_split = (Np - L + 1) / 2;
    if (n_split < L) n_split = L;
    if (n_split > F + L) n_split = F + L;

    for (int s = 0; s < a.n_thr; ++s) {
        const R th = a.thr[(size_t)b * a.n_thr + s];
        // frames 0 .. Q-1 of the extended spectrogram: the left edge pads and the first frame to update
        __syncthreads();
        for (int i = tid; i < Q * Np; i += nthr) S[(size_t)((i / Np) & (NR - 1)) * Np + (i % Np)] = load_state(gS + i);
        for (int i = tid; i < Np; i += nthr) A[((Q - 1) & 1) * Np + i] = gA[(size_t)(Q - 1) * Np + i];
        __syncthreads();
        for (int m = 0; m < T; ++m) {
            const int me = m + Q - 1;                              // extended frame being updated
            float2 *cur = S + (size_t)(me & (NR - 1)) * Np;
            // the next frame, fetched now, stored after this frame's rounds (its slot held frame me + 1 - NR, long final)
            const bool have_next = me + 1 < Tp && m + 1 < T;
            const R *amp_cur = A + (me & 1) * Np;
            float2 nxt[3];
            R anxt[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = tid + j * nthr;
                nxt[j] = (have_next && i < Np) ? load_state(gS + (size_t)(me + 1) * Np + i) : make_float2((R)0, (R)0);
                anxt[j] = (have_next && i < Np) ? gA[(size_t)(me + 1) * Np + i] : (R)0;
            }
            // bins are dealt to lanes (SPLIT: to groups of 8 lanes)
            constexpr int LPB = SPLIT ? 8 : 1;
            const int slot = tid / LPB, nslots = nthr / LPB, jl = tid % LPB;
            auto update = [&](int n) __attribute__((always_inline)) {
                if constexpr (SPLIT) {
                    if (uni) nf_update_split<QT, LT, COMPAT, true>(n, jl, me, th, S, W, Mk, m_uni, amp_cur, cur, Q, L, F, Np, NR, RW);
                    else nf_update_split<QT, LT, COMPAT, false>(n, jl, me, th, S, W, Mk, m_uni, amp_cur, cur, Q, L, F, Np, NR, RW);
                } else {
                    if (uni) nf_update<QT, LT, COMPAT, true, R, float2>(n, me, th, S, W, Mk, m_uni, amp_cur, cur, Q, L, F, Np, NR, RW);
                    else nf_update<QT, LT, COMPAT, false, R, float2>(n, me, th, S, W, Mk, m_uni, amp_cur, cur, Q, L, F, Np, NR, RW);
                }
            };
            if constexpr (COMPAT) {
                for (int n = L + slot; n < n_split; n += nslots) update(n);
                __syncthreads();
                for (int n0 = n_split; n0 < F + L;) {
                    int n1 = (n0 + Np - L + 1) / 2;
                    if (n1 > F + L) n1 = F + L;
                    // a bin n <= 2L writes its image into column 2L - n; keep such bins out of rounds whose other bins could
                    // read that column through the flat offset (only possible for F <= 3L - 1): run them one by one
                    if (n0 <= 2 * L) n1 = n0 + 1;
                    for (int n = n0 + slot; n < n1; n += nslots) update(n);
                    __syncthreads();
                    n0 = n1;
                }
            } else {
                for (int n = L + slot; n < F + L; n += nslots) update(n);
                __syncthreads();
            }
            // frame me is final for this sweep: write it out (pad columns included), bring the next frame into the ring
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = tid + j * nthr;
                if (i < Np) {
                    gS[(size_t)me * Np + i] = cur[i];
                    if (have_next) { S[(size_t)((me + 1) & (NR - 1)) * Np + i] = nxt[j]; A[((me + 1) & 1) * Np + i] = anxt[j]; }
                }
            }
            __syncthreads();
        }
        __threadfence();   // the next sweep reads these frames back from memory
    }
}

template <int QT, int LT, bool COMPAT, bool SPLIT, typename R = float>
hipError_t launch_ts(const NfArgsT<R> &a, int B, int threads, size_t lds, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};   // one bit per device
    int attr_dev;
    if (lws::attr_needed(attr_set, &attr_dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nofuture<QT, LT, COMPAT, SPLIT, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lws::attr_done(attr_set, attr_dev);
    }
    hipLaunchKernelGGL((k_nofuture<QT, LT, COMPAT, SPLIT, R>), dim3(B), dim3(threads), lds, s, a);
    return hipGetLastError();
}
// threads: of the one-lane-per-bin variant; the eight-lane one takes up to 1024
template <int QT, int LT, bool COMPAT>
hipError_t launch_t(const NfArgs &a, int B, int threads, size_t lds, hipStream_t s) {
    const char *ev = getenv("LWS_NOFUTURE_SERIAL_TAPS");   // verification only: one lane sums every tap, in the generic engine's order
    if (ev && ev[0] == '1') return launch_ts<QT, LT, COMPAT, false>(a, B, threads, lds, s);
    int t8 = threads * 4;
    if (t8 > 1024) t8 = 1024;
    return launch_ts<QT, LT, COMPAT, true>(a, B, t8, lds, s);
}

struct NfShape { int NR, threads; size_t lds; bool ok; };

// rows: weight rows the kernel keeps in LDS (Q for a summarised tensor, the row period of a general one; 0: not periodic)
NfShape shape_of(int F, int T, int L, int Q, int Qp, int rows, bool fp64 = false) {
    NfShape sh{0, 0, 0, false};
    if (rows < 1 || (Qp == Q && rows != Q) || (Qp != Q && (Qp != 2 * (F - 1) || Qp % rows != 0)) || Q < 2 || L < 1 || T < 1) return sh;
    const int Np = F + 2 * L;
    int NR = 2;
    while (NR < Q + 1) NR *= 2;          // frames me - Q + 1 .. me and the prefetched me + 1
    sh.NR = NR;
    sh.threads = Np >= 384 ? 512 : (Np >= 192 ? 256 : (Np >= 96 ? 128 : 64));
    // three elements of a frame per thread at most (the prefetch of the next frame): the eight-lanes-per-bin variant runs up to
    // 1024 threads (frames of up to 3072 columns: 4096-point STFTs), the one-lane verification variant `threads`
    {
        const char *ev = getenv("LWS_NOFUTURE_SERIAL_TAPS");
        const int launched = (fp64 || (ev && ev[0] == '1')) ? sh.threads : (4 * sh.threads > 1024 ? 1024 : 4 * sh.threads);
        if (3 * launched < Np) return sh;
    }
    if (Q * (L + 1) > 64) return sh;     // one 64-bit participation mask per weight row
    const size_t es = fp64 ? 2 : 1;      // fp64: twice the bytes per value
    sh.lds = (size_t)NR * Np * 8 * es + (size_t)rows * Q * (L + 1) * 8 * es + (size_t)(rows + 1) * 8 + (size_t)2 * Np * 4 * es + 16;
    if (sh.lds > 160 * 1024) return sh;
    sh.ok = true;
    return sh;
}

}  // namespace

bool nofuture_lds_supports(int F, int T, int L, int Q, int Qp, int rows) { return shape_of(F, T, L, Q, Qp, rows).ok; }
// (summarised tensors only: the rows of a general tensor repeat to 1e-9, not to the bit)
bool nofuture_lds64_supports(int F, int T, int L, int Q, int Qp, int rows) { return Qp == Q && shape_of(F, T, L, Q, Qp, rows, true).ok; }

// Smallest P <= pmax dividing Qp such that the rows of W[Qp][Q][L+1] (complex128 interleaved) repeat with period P -- Q for a
// summarised tensor (trivially), frame / gcd(frame, hop) for create_weights' general ones (lws.pyx:164-181) -- or 0.  The kernels
// above then index row (bin mod P) where the reference indexes row bin (LWSfractionalQ, lwslib.cpp:393,408: mod = bin,
// modneg = N - bin): the same weights to 1e-9 of the largest one (rounded to fp32 afterwards).
int weights_row_period(const double *W, int Qp, int Q, int L, int pmax) {
    if (!W || Qp < 1) return 0;
    const size_t RQ = (size_t)Q * (L + 1);
    double scale = 0;
    for (size_t x = 0; x < (size_t)Qp * RQ; ++x) scale = std::fmax(scale, std::hypot(W[2 * x], W[2 * x + 1]));
    for (int P = 1; P <= pmax && P <= Qp; ++P) {
        if (Qp % P != 0) continue;
        bool ok = true;
        for (int p = P; p < Qp && ok; ++p)
            for (size_t x = 0; x < RQ; ++x) {
                const size_t i = (size_t)p * RQ + x, j = (size_t)(p % P) * RQ + x;
                if (std::hypot(W[2 * i] - W[2 * j], W[2 * i + 1] - W[2 * j + 1]) > 1e-9 * scale) { ok = false; break; }
            }
        if (ok) return P;
    }
    return 0;
}

hipError_t launch_nofuture_lds(const GenericArgs<float> &g, int B, int rows, hipStream_t stream) {
    const NfShape sh = shape_of(g.F, g.T, g.L, g.Q, g.Qp, rows);
    if (!sh.ok) return hipErrorInvalidValue;
    NfArgs a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    a.w = g.w[g.wsel].w; a.flag = g.w[g.wsel].flag;
    a.F = g.F; a.T = g.T; a.L = g.L; a.Q = g.Q; a.n_thr = g.n_thr; a.NR = sh.NR; a.rows = rows;
    a.compat = (g.mode == MODE_NOFUTURE_Q4_COMPAT);
    if (rows != g.Q) return a.compat ? hipErrorInvalidValue : launch_t<0, 0, false>(a, B, sh.threads, sh.lds, stream);   // general weights
    if (a.compat) {
        if (g.Q != 4) return hipErrorInvalidValue;
        return g.L == 5 ? launch_t<4, 5, true>(a, B, sh.threads, sh.lds, stream) : launch_t<4, 0, true>(a, B, sh.threads, sh.lds, stream);
    }
    if (g.Q == 4 && g.L == 5) return launch_t<4, 5, false>(a, B, sh.threads, sh.lds, stream);
    if (g.Q == 2 && g.L == 5) return launch_t<2, 5, false>(a, B, sh.threads, sh.lds, stream);
    if (g.Q == 8 && g.L == 5) return launch_t<8, 5, false>(a, B, sh.threads, sh.lds, stream);
    return launch_t<0, 0, false>(a, B, sh.threads, sh.lds, stream);
}

// The no-future sweeps of an fp64 plan: the one-lane-per-bin variant above in double -- update_bin / update_bin_nfq4 of lws_generic.hip on
// the LDS ring, same arithmetic, same order, no contraction: generic_fp64's results bit for bit (tests/test_gpu_online64.py).
hipError_t launch_nofuture_lds64(const GenericArgs<double> &g, int B, int rows, hipStream_t stream) {
    const NfShape sh = shape_of(g.F, g.T, g.L, g.Q, g.Qp, rows, true);
    if (!sh.ok) return hipErrorInvalidValue;
    NfArgsT<double> a;
    a.state = g.state; a.amp = g.amp; a.thr = g.thr;
    a.w = g.w[g.wsel].w; a.flag = g.w[g.wsel].flag;
    a.F = g.F; a.T = g.T; a.L = g.L; a.Q = g.Q; a.n_thr = g.n_thr; a.NR = sh.NR; a.rows = rows;
    a.compat = (g.mode == MODE_NOFUTURE_Q4_COMPAT);
    if (rows != g.Q || g.Qp != g.Q) return hipErrorInvalidValue;
    if (a.compat) {
        if (g.Q != 4) return hipErrorInvalidValue;
        return g.L == 5 ? launch_ts<4, 5, true, false, double>(a, B, sh.threads, sh.lds, stream) : launch_ts<4, 0, true, false, double>(a, B, sh.threads, sh.lds, stream);
    }
    if (g.Q == 4 && g.L == 5) return launch_ts<4, 5, false, false, double>(a, B, sh.threads, sh.lds, stream);
    if (g.Q == 2 && g.L == 5) return launch_ts<2, 5, false, false, double>(a, B, sh.threads, sh.lds, stream);
    return launch_ts<0, 0, false, false, double>(a, B, sh.threads, sh.lds, stream);
}

}  // namespace lws
