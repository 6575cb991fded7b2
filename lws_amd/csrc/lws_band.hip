// lws_band.hip -- the band engine: batch sweeps (LWSanyQ / LWSfractionalQ, lwslib.cpp:283-467) of the shapes no systolic build takes
// -- 5..8 frames per stencil row above 513 bins (lws(2048,256)), more than 8 frames per row (lws(1024,64)), stencils of half-width
// 6..10, table twiddles on long frames -- in fp32 and in fp64, between the systolic builds (4.9-16 ps per bin and sweep) and the
// order-exact generic engine (88-235 ps).
//
// It is lws_sys64.hip's design with its compile-time choices made at run time (lws_band_core.h has the step and says how): one
// workgroup per spectrogram, a sweep slot = nls / 64 waves whose lanes are nls consecutive frames SKW steps apart, one bin per lane
// and step, a slot's output in an LDS ring of R rows addressed by time, the slot after it LAG steps behind, the first slot fed
// from a time-skewed copy of the state in HBM (coalesced rows, requested two steps ahead), the last one writing it back in place;
// neighbour frames in scatter form, the frame's own taps from two register windows.  One barrier per step.  The geometry -- SKW,
// nls, slots -- is chosen per call (band_plan) from the frame length, Q and what the LDS holds.
//
// The step is checked on the CPU: tests/band_emul.cpp compiles lws_band_core.h with g++ and tests/test_band_model.py compares
// it with the fp64 CPU restatement of the reference (Q = 2..16, L = 3..10, fractional Q, every geometry parameter).
#include "lws_band.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "lws_band_host.h"

namespace lws {
namespace {
using namespace band;

constexpr size_t LDS_BYTES = 160 * 1024;

template <typename real> struct BArgs {
    typename cx<real>::type *G;           // [chunk][rows][nls] time-skewed state
    const real *A;                        // [chunk][rows][nls] target magnitudes
    const real *thr;                      // [chunk][n_thr]
    const real *amax;                     // [chunk] largest target magnitude of each spectrogram (k_band_load)
    const typename cx<real>::type *tab;   // the two tables of lws_band_host.h: [Q][LT+1] weights, then [Pt][Q-1] twiddles
    long g_stride;                        // rows * nls
    int n_thr, thr0, ns, nsl;             // this pass: sweeps thr0 .. thr0 + ns - 1 on the first ns of the nsl slots launched
    Geom g;
};

// LDS writes of this step complete, then everybody meets.  (Not __syncthreads(): that waits for the global prefetches too.)
#define BAND_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// One wave's share of a pass.  NH > 0: the slot's frame offsets are shared out between a main wave and NH helper waves a step ahead of it
// (lws_band_core.h: Lane, Split); `part` = 0 main, 1 .. NH the helpers.
template <typename real, int LT, int QT, bool FIRST, bool EXACT, bool HELP, int RLO, int RHI, int NHELP>
__device__ __forceinline__ void band_wave(const BArgs<real> &a, typename cx<real>::type *ring, const typename cx<real>::type *wt, const typename cx<real>::type *tw,
                                          typename cx<real>::type *mail, int nh, int h, typename cx<real>::type *G, const real *A, int s, int lane) {
    using C = typename cx<real>::type;
    Env<real, C> e;
    e.g = a.g;
    e.ring_own = ring + (size_t)s * a.g.R * a.g.nls;
    e.ring_prev = ring + (size_t)(s > 0 ? s - 1 : 0) * a.g.R * a.g.nls;
    e.tw = tw;
    e.wt = wt;
    e.mail = mail;
    e.mail_nh = nh;
    e.mail_h = h;
    e.G = G;
    e.A = A;
    const bool live = s < a.ns;
    e.thr = a.thr[(size_t)blockIdx.x * a.n_thr + a.thr0 + (live ? s : 0)];
    e.last = s == a.ns - 1;
    Lane<real, C, LT, QT, FIRST, EXACT, HELP, RLO, RHI, NHELP> ln(e, lane, s);
    // a slot's main wave starts LAG steps after the slot before it -- two steps later still when helpers run a step ahead of it
    constexpr int SHIFT = HELP ? 1 : (NHELP ? 2 : 0);
    const int t_end = a.g.U + a.g.LAG * (a.ns - 1) + (nh ? 2 : 0);   // U and LAG are even
    int ph = 0;
    for (int t0 = 0; t0 < t_end; t0 += 2) {
        const int u0 = t0 - a.g.LAG * s - SHIFT;      // frame-time of this pair's first step (odd for a helper)
        if (live && u0 >= 0 && u0 < a.g.U) {
            if (u0 == 0) ln.prologue();
            ln.template step<(SHIFT & 1)>(u0, ph);
            ph = ph + 1 == a.g.SKW ? 0 : ph + 1;
        }
        BAND_BARRIER();
        const int u1 = u0 + 1;
        if (live && u1 >= 0 && u1 < a.g.U) {
            if (u1 == 0) ln.prologue();
            ln.template step<((SHIFT + 1) & 1)>(u1, ph);
            ph = ph + 1 == a.g.SKW ? 0 : ph + 1;
        }
        BAND_BARRIER();
    }
}
template <typename real, int LT, int QT, bool FIRST, int NH, int I>
__device__ __forceinline__ void band_helper(const BArgs<real> &a, typename cx<real>::type *ring, const typename cx<real>::type *wt, const typename cx<real>::type *tw,
                                            typename cx<real>::type *mail, int sub, typename cx<real>::type *G, const real *A, int s, int lane) {
    if constexpr (I <= NH) {
        if (sub == I) band_wave<real, LT, QT, FIRST, true, true, Split<QT>::lo(I), Split<QT>::hi(I), 0>(a, ring, wt, tw, mail, NH, I - 1, G, A, s, lane);
        else band_helper<real, LT, QT, FIRST, NH, I + 1>(a, ring, wt, tw, mail, sub, G, A, s, lane);
    }
}

template <typename real, int LT, int QT, bool EXACT, int NH, int MAXT>
__global__ void __launch_bounds__(MAXT) k_band(BArgs<real> a) {
    using C = typename cx<real>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char band_lds[];
    C *ring = reinterpret_cast<C *>(band_lds);
    const int nls = a.g.nls, wps = nls >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves in the order [role][slot][part of the ring row]: the hardware deals consecutive waves to the four SIMDs in turn, so a
    // SIMD gets a main wave AND a helper wave (their instruction counts differ) rather than two of a kind
    const int sub = wave / (a.nsl * wps);                           // 0: the slots' main waves; 1 .. NH: their helpers
    // (... and the helpers of the slots in an order rotated by half the slots against the main waves': the first slot's waves also
    //  issue the loads from the skewed state -- a SIMD gets one of them, not both)
    const int s = ((wave / wps) % a.nsl + (sub & 1) * (a.nsl / 2)) % a.nsl;   // sweep slot
    const int lane = (wave % wps) * 64 + (threadIdx.x & 63);        // place in the slot's ring row
    {
        // a pass none of whose sweeps has a bin above its threshold changes nothing (lwslib.cpp:295-296: strict '>'): the first ~38 of
        // the reference's default 100 sweeps are such sweeps.  The same answer in every thread of the workgroup.
        const real top = a.amax[blockIdx.x];
        bool any = false;
        for (int i = 0; i < a.ns; ++i) any |= top > a.thr[(size_t)blockIdx.x * a.n_thr + a.thr0 + i];
        if (!any) return;
    }
    const int nring = a.nsl * a.g.R * nls;
    // behind the rings: the weights (read at the same address by every lane: a broadcast; as kernel arguments or behind a global
    // pointer the compiler keeps all of them live -- 94 scalar or vector registers for Q = 8), the twiddles, the helpers' mailboxes
    C *wt = ring + nring, *tw = wt + a.g.Q * (LT + 1), *mail = tw + a.g.Pt * (a.g.Q - 1);
    {
        C z; z.x = 0; z.y = 0;
        for (int i = threadIdx.x; i < nring; i += blockDim.x) ring[i] = z;
        for (int i = threadIdx.x; i < a.g.Q * (LT + 1) + a.g.Pt * (a.g.Q - 1); i += blockDim.x) wt[i] = a.tab[i];
        for (int i = threadIdx.x; i < a.nsl * 2 * NH * nls * 2; i += blockDim.x) mail[i] = z;
        __syncthreads();
    }
    C *G = a.G + (size_t)blockIdx.x * a.g_stride;
    const real *A = a.A + (size_t)blockIdx.x * a.g_stride;
    mail += (size_t)s * 2 * NH * nls * 2;
    if constexpr (NH == 0) {
        if (s == 0) band_wave<real, LT, QT, true, EXACT, false, 0, QT - 1, 0>(a, ring, wt, tw, mail, 0, 0, G, A, s, lane);
        else band_wave<real, LT, QT, false, EXACT, false, 0, QT - 1, 0>(a, ring, wt, tw, mail, 0, 0, G, A, s, lane);
    } else {
        if (sub == 0) {
            if (s == 0) band_wave<real, LT, QT, true, true, false, Split<QT>::lo(0), Split<QT>::hi(0), NH>(a, ring, wt, tw, mail, NH, 0, G, A, s, lane);
            else band_wave<real, LT, QT, false, true, false, Split<QT>::lo(0), Split<QT>::hi(0), NH>(a, ring, wt, tw, mail, NH, 0, G, A, s, lane);
        } else {
            if (s == 0) band_helper<real, LT, QT, true, NH, 1>(a, ring, wt, tw, mail, sub, G, A, s, lane);
            else band_helper<real, LT, QT, false, NH, 1>(a, ring, wt, tw, mail, sub, G, A, s, lane);
        }
    }
}

// extended buffers [B][Tp][F + 2 L] <-> the skewed layout (bins 0 .. F-1 and LT images above Nyquist per frame)
template <typename real>
__global__ void __launch_bounds__(256) k_band_load(const typename cx<real>::type *state, const real *amp, typename cx<real>::type *G, real *A,
                                                    real *amax, int L, int LT, int Tp, Geom g, long g_stride) {
    using C = typename cx<real>::type;
    const int bb = blockIdx.y, me = blockIdx.x, F = g.F, Np = F + 2 * L;
    const int j = me & (g.nls - 1), blk = me / g.nls;
    const long base = (long)g.SKW * j + (long)g.P * blk + LT;
    const C *src = state + ((size_t)bb * Tp + me) * Np + L;
    const real *asrc = amp + ((size_t)bb * Tp + me) * Np + L;
    real top = 0;
    for (int b = threadIdx.x; b < F + LT; b += blockDim.x) {
        const int q = b < F ? b : 2 * (F - 1) - b;
        C v = src[q];
        if (b >= F) v.y = -v.y;
        G[(size_t)bb * g_stride + (base + b) * g.nls + j] = v;
        const real am = asrc[q];
        A[(size_t)bb * g_stride + (base + b) * g.nls + j] = am;
        top = am > top ? am : top;
    }
    // the spectrogram's largest target magnitude (non-negative values order like their bit patterns)
    __shared__ real red[256];
    red[threadIdx.x] = top;
    __syncthreads();
    for (int h = blockDim.x / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + h] ? red[threadIdx.x] : red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0 && me >= g.Q - 1 && me < g.T + g.Q - 1) {
        if constexpr (sizeof(real) == 4) atomicMax(reinterpret_cast<unsigned int *>(amax + bb), __float_as_uint((float)red[0]));
        else atomicMax(reinterpret_cast<unsigned long long *>(amax + bb), (unsigned long long)__double_as_longlong((double)red[0]));
    }
}
template <typename real>
__global__ void __launch_bounds__(256) k_band_store(typename cx<real>::type *state, const typename cx<real>::type *G, int L, int LT, int Tp,
                                                     Geom g, long g_stride) {
    using C = typename cx<real>::type;
    const int bb = blockIdx.y, me = blockIdx.x, F = g.F, Np = F + 2 * L;
    const int j = me & (g.nls - 1), blk = me / g.nls;
    const long base = (long)g.SKW * j + (long)g.P * blk + LT;
    C *dst = state + ((size_t)bb * Tp + me) * Np + L;
    for (int b = threadIdx.x; b < F; b += blockDim.x) {
        const C v = G[(size_t)bb * g_stride + (base + b) * g.nls + j];
        dst[b] = v;
        C vc; vc.x = v.x; vc.y = -v.y;          // the Hermitian images in the pad columns (lwslib.cpp:362-367)
        if (b >= 1 && b <= L) dst[-b] = vc;
        if (b >= F - 1 - L && b <= F - 2) dst[2 * (F - 1) - b] = vc;
    }
}

template <typename real, int LT, int QT, bool EXACT, int NH, int MAXT>
hipError_t launch_pass(const BArgs<real> &a, int B, hipStream_t stream) {
    using C = typename cx<real>::type;
    static std::atomic<unsigned long long> done{0};
    const size_t lds = (size_t)a.nsl * (ring_bytes(a.g, sizeof(C)) + mail_bytes(a.g, NH, sizeof(C))) + table_bytes(a.g, LT, sizeof(C));
    int dev = 0;
    if (attr_needed(done, &dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_band<real, LT, QT, EXACT, NH, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done(done, dev);
    }
    k_band<real, LT, QT, EXACT, NH, MAXT><<<dim3(B), dim3(a.g.nls * a.nsl * (1 + NH)), lds, stream>>>(a);
    return hipGetLastError();
}
// The instantiations.  For the Q each family is mostly used with an EXACT one (no test on the frame offsets: a step is straight-line
// code) whose slots are a main wave and helper waves (bp.helpers > 0: two waves per SIMD, 256 vector registers each); one with the
// tests, one wave per slot, for every other Q.
template <typename real>
hipError_t launch_pass_any(const BandPlan &bp, const BArgs<real> &a, int B, hipStream_t stream) {
    const int Q = a.g.Q;
    constexpr bool F32 = std::is_same<real, float>::value;
    constexpr int M8 = F32 ? 512 : 256;
    if (bp.LT == 5) {
        if (Q == 8 && bp.helpers == 1) return launch_pass<real, 5, 8, true, 1, M8>(a, B, stream);   // (fp64: 512 registers a wave, so 256 threads)
        if (Q == 8 && a.g.nls * a.nsl <= 256) return launch_pass<real, 5, 8, true, 0, 256>(a, B, stream);   // (straight-line: one wave per SIMD's registers)
        if (Q <= 8) return launch_pass<real, 5, 8, false, 0, M8>(a, B, stream);
        if constexpr (F32) {
            if (Q == 16 && bp.helpers == 3) return launch_pass<real, 5, 16, true, 3, 512>(a, B, stream);
            if (Q == 16) return launch_pass<real, 5, 16, true, 0, 256>(a, B, stream);
        }
        return launch_pass<real, 5, 16, false, 0, 256>(a, B, stream);
    }
    if (bp.LT == 8) {   // (Q = 4 only: band_plan)
        if constexpr (F32) {
            if (bp.helpers == 1) return launch_pass<real, 8, 4, true, 1, 512>(a, B, stream);
        }
        return launch_pass<real, 8, 4, true, 0, 256>(a, B, stream);
    }
    if constexpr (F32) {
        if (Q == 4 && bp.helpers == 1) return launch_pass<real, 10, 4, true, 1, 512>(a, B, stream);
    }
    if (Q == 4) return launch_pass<real, 10, 4, true, 0, 256>(a, B, stream);
    if (Q <= 8) return launch_pass<real, 10, 8, false, 0, 256>(a, B, stream);
    return launch_pass<real, 10, 16, false, 0, 256>(a, B, stream);
}
// threads a workgroup may have (the instantiation's launch bound) and the helper waves per slot of the build a plan runs on
inline int helpers_of(bool fp64, int LT, int Q) {
    if (LT == 5 && Q == 8) return Split<8>::NH;            // (fp64 too: a ring of 513 bins leaves one slot -- a second wave per CU)
    if (LT == 5 && Q == 16 && !fp64) return Split<16>::NH;
    if ((LT == 10 || LT == 8) && Q == 4 && !fp64) return Split<4>::NH;  // (fp64: the slots the LDS holds already fill the 256 threads its registers allow)
    return 0;
}
inline int max_threads(bool fp64, int LT, int QT, int helpers) { return fp64 ? 256 : ((helpers || (LT == 5 && QT == 8)) ? 512 : 256); }

int env_i(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Every row of W the twiddle image of row 0 to `tol` of the largest weight?  (weights_twiddle accepts 1e-9: enough to choose an
// engine whose arithmetic is fp32; an fp64 plan promises the reference's values to rounding.)
bool rows_are_twiddles(const double *W, int Q, int Qp, int L, int Pt, int s, double tol) {
    const int K1 = L + 1;
    double scale = 0;
    for (size_t x = 0; x < (size_t)Qp * Q * K1; ++x) scale = std::max(scale, std::hypot(W[2 * x], W[2 * x + 1]));
    auto at = [&](int p, int r, int k, int c) { return W[2 * (((size_t)p * Q + r) * K1 + k) + c]; };
    for (int p = 0; p < Qp; ++p)
        for (int r = 0; r < Q; ++r) {
            double cr, ci;
            unit((long long)p * r * s, Pt, &cr, &ci);
            for (int k = 0; k < K1; ++k) {
                if (r == 0 && k == 0) continue;
                const double br = at(0, r, k, 0), bi = at(0, r, k, 1);
                if (std::hypot(at(p, r, k, 0) - (br * cr - bi * ci), at(p, r, k, 1) - (br * ci + bi * cr)) > tol * scale) return false;
                // the reference skips a weight by its own magnitude (lws.pyx:232); rows that disagree about that cannot share row 0
                if ((std::hypot(at(p, r, k, 0), at(p, r, k, 1)) > 1e-12) != (std::hypot(br, bi) > 1e-12)) return false;
            }
        }
    return true;
}

// relative time of a step by the waves that share a SIMD (the headline kernel's measurement: two waves stretch a step by 1.4)
inline double step_cost(int waves) {
    const int k = (waves + 3) / 4;
    return k <= 1 ? 1.0 : 1.0 + 0.45 * (k - 1);
}

}  // namespace

bool band_plan(bool fp64, int B, int F, int T, int L, int Q, int Qp, int update, int n_thr, const double *W, BandPlan *out) {
    if (!W || update != 2 || T < 1 || n_thr < 1 || Q < 2 || Q > 16 || L < 1 || L > 10 || Qp < 1) return false;
    // the stencil half-width the kernel is compiled for: 5, 10, and 8 for Q = 4 (`lws(1024,256,L=8)`: frames 10 steps apart instead of 12)
    const int LT = L <= 5 ? 5 : ((L <= 8 && Q == 4) ? 8 : 10), QT = Q <= 8 ? 8 : 16;
    if (F < 2 * LT + 7) return false;
    int Pt = 0, s = 0;
    // (any twiddle period whose table still leaves room for a ring: a hop with no common factor with the frame has Pt = the frame)
    if (!weights_twiddle(W, Q, Qp, L, 4096, &Pt, &s)) return false;
    if (Pt < 1) { Pt = 1; s = 0; }
    // (fp64: the rows must be the twiddle images of row 0 to rounding, or the results would not be the reference's.  The general
    //  tensors create_weights builds for a hop that does not divide the frame -- one row per bin, numpy's exp of an angle of up to N
    //  turns -- are 1e-13 to 1e-16 of a turn off, which a few sweeps amplify a thousandfold: fp64 plans with such tensors stay on the
    //  order-exact engine, whatever the check below would say)
    if (fp64 && Qp != Q) return false;
    if (!rows_are_twiddles(W, Q, Qp, L, Pt, s, fp64 ? 1e-13 : 1e-9)) return false;
    const size_t csize = fp64 ? 16 : 8;
    // (LWS_BAND_NO_HELPERS=1: the exact builds' one-wave-per-slot variant -- comparison runs)
    const int helpers = env_i("LWS_BAND_NO_HELPERS", 0) ? 0 : helpers_of(fp64, LT, Q);
    const int maxt = max_threads(fp64, LT, QT, helpers);
    BandPlan best{};
    double best_cost = 1e300;
    const int skw_force = env_i("LWS_BAND_SKW", 0), nls_force = env_i("LWS_BAND_NLS", 0), ns_force = env_i("LWS_BAND_NS", 0);   // (tests)
    for (int nls = 64; nls * (1 + helpers) <= maxt; nls *= 2) {
        if (nls_force && nls != nls_force) continue;
        for (int SKW = LT + 2; SKW <= LT + 2 + 10; ++SKW) {
            if (skw_force && SKW != skw_force) continue;
            const Geom g = geometry(F, T, Q, LT, SKW, nls, Pt, helpers);
            if (table_bytes(g, LT, csize) + 1024 >= LDS_BYTES) continue;
            const size_t avail = LDS_BYTES - table_bytes(g, LT, csize);
            int NS = (int)std::min<size_t>(avail / (ring_bytes(g, csize) + mail_bytes(g, helpers, csize)), (size_t)(maxt / (nls * (1 + helpers))));
            NS = std::min(NS, n_thr);
            if (ns_force) NS = std::min(NS, ns_force);
            if (NS < 1) continue;
            // steps per sweep (the passes of a call: full ones, then the rest), by what a step costs with that many waves
            const int full = n_thr / NS, rest = n_thr % NS;
            const double steps = (double)full * (g.U + (double)g.LAG * (NS - 1)) + (rest ? g.U + (double)g.LAG * (rest - 1) : 0.0);
            const double cost = steps * step_cost(NS * nls / 64);      // (helper waves share their main wave's work: not counted)
            if (cost < best_cost) {
                best_cost = cost;
                best.g = g; best.NS = NS;
            }
        }
    }
    if (best_cost >= 1e300) return false;
    best.LT = LT; best.QT = QT; best.Pt = Pt; best.s = s; best.L = L; best.fp64 = fp64; best.helpers = helpers;
    // spectrograms that go through the skewed scratch at a time: at most 32 GiB of it (LWS_BAND_CHUNK: for tests)
    const size_t per = (size_t)best.g.rows * best.g.nls * (csize + csize / 2);
    int chunk = (int)std::min<size_t>((size_t)std::max(B, 1), std::max<size_t>(1, ((size_t)32 << 30) / per));
    const int cf = env_i("LWS_BAND_CHUNK", 0);
    if (cf > 0) chunk = std::min(chunk, cf);
    best.chunk = chunk;
    best.state_bytes = (size_t)chunk * best.g.rows * best.g.nls * csize;
    best.amp_bytes = (size_t)chunk * best.g.rows * best.g.nls * (csize / 2) + (size_t)chunk * (csize / 2);   // (+ a largest magnitude per spectrogram)
    if (out) *out = best;
    return true;
}

const char *band_name(const BandPlan &bp) { return bp.fp64 ? "band_fp64" : "band_fp32"; }

std::vector<unsigned char> band_tables(const BandPlan &bp, const double *W_host) {
    std::vector<double> wtd, twd;
    tables(W_host, bp.g.Q, bp.L, bp.LT, bp.Pt, bp.s, wtd, twd);
    wtd.insert(wtd.end(), twd.begin(), twd.end());
    std::vector<unsigned char> out(wtd.size() * (bp.fp64 ? sizeof(double) : sizeof(float)));
    if (bp.fp64) memcpy(out.data(), wtd.data(), out.size());
    else for (size_t i = 0; i < wtd.size(); ++i) reinterpret_cast<float *>(out.data())[i] = (float)wtd[i];
    return out;
}

template <typename real>
hipError_t launch_band(const BandPlan &bp, const GenericArgs<real> &ga, const void *tables_dev, int B, void *gs, void *gamp, hipStream_t stream,
                       int *launches, hipEvent_t ev0, hipEvent_t ev1) {
    using C = typename cx<real>::type;
    if (B <= 0 || ga.n_thr <= 0) return hipSuccess;
    if (ga.mode != MODE_BATCH || ga.L != bp.L || ga.F != bp.g.F || ga.T != bp.g.T || ga.Q != bp.g.Q || !tables_dev) return hipErrorInvalidValue;
    const Geom &g = bp.g;
    const int Tp = g.T + 2 * (g.Q - 1);
    const size_t Np = g.F + 2 * bp.L;
    C *G = static_cast<C *>(gs);
    real *A = static_cast<real *>(gamp);
    const long g_stride = g.rows * g.nls;
    hipError_t e;
    if (ev0) (void)hipEventRecord(ev0, stream);
    int n_all = 0;
    for (int b0 = 0; b0 < B; b0 += bp.chunk) {
        const int Bc = std::min(bp.chunk, B - b0);
        // rows no frame owns are read by lanes whose results are discarded, and must be zeros for the lanes that do use them
        if ((e = hipMemsetAsync(G, 0, (size_t)Bc * g_stride * sizeof(C), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(A, 0, (size_t)Bc * g_stride * sizeof(real), stream)) != hipSuccess) return e;
        real *amax = A + (size_t)bp.chunk * g_stride;      // behind the magnitudes
        if ((e = hipMemsetAsync(amax, 0, (size_t)Bc * sizeof(real), stream)) != hipSuccess) return e;
        C *state = ga.state + (size_t)b0 * Tp * Np;
        k_band_load<real><<<dim3(Tp, Bc), 256, 0, stream>>>(state, ga.amp + (size_t)b0 * Tp * Np, G, A, amax, bp.L, bp.LT, Tp, g, g_stride);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        BArgs<real> a;
        a.G = G; a.A = A; a.thr = ga.thr + (size_t)b0 * ga.n_thr; a.amax = amax;
        a.tab = static_cast<const C *>(tables_dev);
        a.g_stride = g_stride; a.n_thr = ga.n_thr; a.nsl = bp.NS; a.g = g;
        for (int i0 = 0; i0 < ga.n_thr; i0 += bp.NS, ++n_all) {
            a.thr0 = i0;
            a.ns = std::min(bp.NS, ga.n_thr - i0);
            if ((e = launch_pass_any<real>(bp, a, Bc, stream)) != hipSuccess) return e;
        }
        k_band_store<real><<<dim3(Tp, Bc), 256, 0, stream>>>(state, G, bp.L, bp.LT, Tp, g, g_stride);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    if (launches) *launches = n_all;
    return hipSuccess;
}
template hipError_t launch_band<float>(const BandPlan &, const GenericArgs<float> &, const void *, int, void *, void *, hipStream_t, int *, hipEvent_t, hipEvent_t);
template hipError_t launch_band<double>(const BandPlan &, const GenericArgs<double> &, const void *, int, void *, void *, hipStream_t, int *, hipEvent_t, hipEvent_t);

}  // namespace lws
