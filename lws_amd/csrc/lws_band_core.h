// lws_band_core.h -- the per-lane step of the band engine (lws_band.hip).
//
// The same text compiles for the GPU (hipcc: every function __device__) and for a host-side emulation (g++:
// tests/band_emul.cpp steps through the schedule lane by lane on the CPU and is compared with the fp64 CPU restatement of the reference by
// tests/test_band_model.py), so the schedule -- ring ages, Hermitian images, frame wrap-around -- is debugged without a GPU.
//
// What a step computes is one bin of LWSanyQ / LWSfractionalQ (lwslib.cpp:283-467) per lane; the schedule is the fp64 systolic
// engine's (lws_sys64.hip: lane = frame, frames SKW steps apart, one bin per step, a slot's output in an LDS ring addressed by
// time, neighbour frames in scatter form), with everything that engine fixes at compile time -- Q, the quarter-turn twiddles of
// the weight rows, the phase of a step in an eight-times unrolled loop -- taken at run time:
//   * any Q up to QT (the frame offsets r = 1..Q-1 are blocks of straight-line code behind a wave-uniform test);
//   * any stencil half-width up to LT (a narrower stencil runs with zero weights in the outer columns);
//   * any weight tensor with create_weights' twiddle structure W[p][r][k] = W[0][r][k] tau_r^p, tau_r = exp(2 pi j r s / Pt)
//     (lws.pyx:160-181): the value a step receives from the frames r apart -- position w of frame m - r and of frame m + r --
//     is turned by tau_r^w once, after which the weight of target bin w -+ k is V[r][k] = W[0][r][k] tau_r^k whatever the bin:
//         W[(w+k) mod][r][k] A + conj(.) B  =  V (tau^w A) + conj(V) (conj(tau^w) B)
//         W[-(w-k) mod][r][k] B + conj(.) A =  V (conj(tau^w) B) + conj(V) (tau^w A)
//     (the two groupings of lwslib.cpp:333-352), so the table of twiddles is indexed by the lane's position, the weights by
//     nothing but (r, k), and both are the same for every lane.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BAND_FN __device__ __forceinline__
#else
#include <cassert>
#include <cmath>
#define BAND_FN inline
#endif

namespace lws {
namespace band {

constexpr int PFD = 2;   // steps a load from the skewed state is issued ahead of its use (the step loop is unrolled by this)

// The geometry of a call (host: band_geometry() in lws_band.hip; tests/band_emul.cpp uses the same function).
struct Geom {
    int F, T, Q;          // bins, frames, frame offsets of the stencil (0 .. Q-1)
    int SKW, nls;         // steps between consecutive frames; frames in flight per sweep slot = lanes of a ring row
    int P, gap;           // steps a lane spends on a frame (a multiple of SKW, >= F + LT and >= nls SKW); P - nls SKW
    int LAG, R;           // steps between consecutive sweep slots; rows of a slot's ring
    int nblk, U;          // blocks of nls frames; steps of one sweep
    int Pt;               // period of the weights' twiddle, in bins
    int lg;               // log2(nls)
    long rows;            // rows of the skewed state of one spectrogram
};

template <typename real> BAND_FN real fma_(real a, real b, real c);
template <> BAND_FN float fma_<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> BAND_FN double fma_<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
#if defined(__HIPCC__)
BAND_FN float rsqrt_(float x) { return __frsqrt_rn(x); }
BAND_FN double rsqrt_(double x) { return rsqrt(x); }
// a pair of reals that the fp32 build keeps in an aligned register pair: its arithmetic is one packed instruction per pair
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; a scalar broadcast into both halves and a negation are operand modifiers)
template <typename real> struct pair_of { typedef real type __attribute__((ext_vector_type(2))); };
template <typename real> using V2 = typename pair_of<real>::type;
template <typename real> BAND_FN V2<real> vfma(V2<real> a, V2<real> b, V2<real> c) { return __builtin_elementwise_fma(a, b, c); }
#else
inline float rsqrt_(float x) { return 1.0f / std::sqrt(x); }
inline double rsqrt_(double x) { return 1.0 / std::sqrt(x); }
template <typename real> struct V2 {
    real x, y;
    V2 operator+(V2 o) const { return V2{x + o.x, y + o.y}; }
    V2 operator-(V2 o) const { return V2{x - o.x, y - o.y}; }
    V2 operator*(V2 o) const { return V2{x * o.x, y * o.y}; }
    V2 operator-() const { return V2{-x, -y}; }
};
template <typename real> inline V2<real> vfma(V2<real> a, V2<real> b, V2<real> c) { return V2<real>{fma_<real>(a.x, b.x, c.x), fma_<real>(a.y, b.y, c.y)}; }
#endif
template <typename real> BAND_FN V2<real> splat(real v) { return V2<real>{v, v}; }
template <typename real> BAND_FN V2<real> turn(V2<real> v) { return V2<real>{-v.y, v.x}; }   // j v
template <typename real, typename C> BAND_FN V2<real> pr(C v) { return V2<real>{v.x, v.y}; }
template <typename real> BAND_FN V2<real> vsel(bool c, V2<real> a, V2<real> b) { return V2<real>{c ? a.x : b.x, c ? a.y : b.y}; }

// Pair arithmetic with one component of a complex weight w = (w.x, w.y) as the scalar factor.  Generic text first; on the GPU in fp32
// each is ONE packed instruction whose operand modifiers pick the component for both halves, swap the halves of the pair (j s) and
// negate -- written out below because the compiler does not derive them (it builds (w.x, w.x) with two moves per weight).
template <typename real, typename C> BAND_FN V2<real> mul_x(C w, V2<real> s) { return splat<real>(w.x) * s; }                     // w.x s
template <typename real, typename C> BAND_FN V2<real> mul_xj(C w, V2<real> s) { return splat<real>(w.x) * turn<real>(s); }          // w.x j s
template <typename real, typename C> BAND_FN V2<real> fma_x(V2<real> a, C w, V2<real> s) { return vfma<real>(splat<real>(w.x), s, a); }    // a + w.x s
template <typename real, typename C> BAND_FN V2<real> fma_y(V2<real> a, C w, V2<real> s) { return vfma<real>(splat<real>(w.y), s, a); }    // a + w.y s
template <typename real, typename C> BAND_FN V2<real> fnma_y(V2<real> a, C w, V2<real> s) { return vfma<real>(-splat<real>(w.y), s, a); }  // a - w.y s
template <typename real, typename C> BAND_FN V2<real> fma_yj(V2<real> a, C w, V2<real> s) { return vfma<real>(splat<real>(w.y), turn<real>(s), a); }   // a + w.y j s
#if defined(__HIPCC__)
typedef float band_v2f __attribute__((ext_vector_type(2)));
template <> BAND_FN band_v2f mul_x<float, float2>(float2 w, band_v2f s) {
    band_v2f r; const band_v2f wv = {w.x, w.y};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(wv), "v"(s));
    return r;
}
template <> BAND_FN band_v2f mul_xj<float, float2>(float2 w, band_v2f s) {
    band_v2f r; const band_v2f wv = {w.x, w.y};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(wv), "v"(s));
    return r;
}
template <> BAND_FN band_v2f fma_x<float, float2>(band_v2f a, float2 w, band_v2f s) {
    const band_v2f wv = {w.x, w.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(a) : "v"(wv), "v"(s));
    return a;
}
template <> BAND_FN band_v2f fma_y<float, float2>(band_v2f a, float2 w, band_v2f s) {
    const band_v2f wv = {w.x, w.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(wv), "v"(s));
    return a;
}
template <> BAND_FN band_v2f fnma_y<float, float2>(band_v2f a, float2 w, band_v2f s) {
    const band_v2f wv = {w.x, w.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(a) : "v"(wv), "v"(s));
    return a;
}
template <> BAND_FN band_v2f fma_yj<float, float2>(band_v2f a, float2 w, band_v2f s) {
    const band_v2f wv = {w.x, w.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a) : "v"(wv), "v"(s));
    return a;
}
#endif

template <typename C> BAND_FN C cj(C v) { v.y = -v.y; return v; }
template <typename C> BAND_FN C sel(bool c, C a, C b) { C r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; return r; }

// What a wave (a sweep slot, or one of the waves a slot is spread over) needs besides its lanes' state.
template <typename real, typename C> struct Env {
    Geom g;
    C *ring_own;            // LDS: this slot's output, R rows of nls lanes, row = time mod R
    const C *ring_prev;     // LDS: the output of the slot before (unused by the first slot of a pass)
    const C *tw;            // LDS: [Pt][Q-1] twiddles tau_r^p
    const C *wt;            // LDS: [Q][LT+1]: row 0 the frame's own taps W[0][0][k], row r the neighbour weights V[r][k] (the same address in every lane)
    C *G;                   // the skewed state of this spectrogram: frame me, bin b at row SKW (me % nls) + P (me / nls) + b + LT
    const real *A;          // target magnitudes, same addressing (row u holds the bin a lane completes at frame-time u, row u + LT the one it receives)
    C *mail;                // LDS: the mailbox of this slot's helper waves, [2 (parity of the step)][helpers][nls][2] (see Lane)
    int mail_nh, mail_h;    // helpers of the slot; which of them this wave is
    real thr;
    bool last;              // this slot's output is what the pass leaves in the skewed state
};

// How the frame offsets of an exact build are shared out between the main wave and its helpers (see Lane): block r (0-based; frame
// offset r + 1) belongs to part i if b[i] <= r < b[i + 1]; part 0 is the main wave's.
template <int QT> struct Split { static constexpr int NH = 0; };
template <> struct Split<4> { static constexpr int NH = 1; static constexpr int lo(int i) { return i == 0 ? 0 : 1; } static constexpr int hi(int i) { return i == 0 ? 1 : 3; } };
template <> struct Split<8> { static constexpr int NH = 1; static constexpr int lo(int i) { return i == 0 ? 0 : 3; } static constexpr int hi(int i) { return i == 0 ? 3 : 7; } };
template <> struct Split<16> { static constexpr int NH = 3; static constexpr int lo(int i) { return i == 0 ? 0 : 4 * i - 1; } static constexpr int hi(int i) { return 4 * i + 3; } };

// EXACT: Q == QT -- the tests on the frame offsets are decided at compile time and a step is straight-line code.
//
// Helper waves (exact builds).  A sweep slot's ring fills most of the LDS, so a slot that is ONE wave per 64 frames leaves a wave alone
// on its SIMD with every LDS round trip exposed.  With NHELP > 0 the frame offsets are shared out: the MAIN wave (HELP = false) takes
// offsets RLO+1 .. RHI, the frame's own taps, the re-projection and all stores; each helper wave (HELP = true) takes a range of the far
// offsets for the same lanes ONE STEP AHEAD of the main wave -- it reads what the main wave will receive a step later (every such value
// is at least two steps old for offsets >= 2) -- keeps sums of its own for bins c .. c + 2 LT and leaves what the bin it completes
// gets from its offsets, and their share of the DC / Nyquist imaginary part, in a mailbox (LDS, two cells deep by the step's parity)
// that the main wave adds to its own a step later.  The order of a bin's sum changes (per helper a partial sum): rounding only.
template <typename real, typename C, int LT, int QT, bool FIRST, bool EXACT = false, bool HELP = false, int RLO = 0, int RHI = QT - 1, int NHELP = 0>
struct Lane {
    static constexpr int NA = 2 * LT + 1, NRT = QT - 1, K1 = LT + 1, NB = RHI - RLO;
    static constexpr int LEAD = HELP ? 1 : 0;           // steps this wave runs ahead of the slot's main wave
    static_assert(RLO >= 0 && RHI <= NRT && NB >= 1 && (!HELP || (EXACT && RLO >= 1)) && (NHELP == 0 || (EXACT && !HELP)), "roles");
    using P = V2<real>;
    // what a step receives from the frames r apart -- position w of frame me - r (L) and of frame me + r (R), tau_r^w (T) -- and what
    // it makes of it: with A' = tau A, B' = conj(tau) B the sum S = A' + B' and D = j (A' - B'), so that a weight v adds
    // v.x S +- v.y D to a bin (the grouped form of lwslib.cpp:310-311; + for the taps below a bin, - above)
    struct In { C L, R, T; };
    struct SD { P S, D; };
    // lane state
    P acc[NA];               // sums of bins c .. c + 2 LT (of this wave's frame offsets)
    C cn[HELP ? 1 : LT + 1]; // cn[k]: new value of bin c - k (below DC: the image, lwslib.cpp:362-364, as it stands at that moment)
    C co[HELP ? 1 : LT + 1]; // co[k]: old value of bin c + k
    real yE;                 // DC / Nyquist: the imaginary part of the bin's sum (see step)
    In nx;                   // inputs of the next step's first frame offset, requested from the LDS one step early
    C nxO, nxI;              // ... its old value of the frame itself and the image above Nyquist it may have to write
    C pfO[FIRST && !HELP ? PFD : 1], pfR[FIRST ? PFD : 1][FIRST ? NB : 1];   // first slot: inputs of the next PFD steps from the skewed state
    real pfA[HELP ? 1 : PFD];
    C wfirst[K1];            // the weights of this wave's first frame offset, and of the frame's own taps (main wave): constants of the
    C wown[HELP ? 1 : LT + 1];   // call, kept in registers -- read at the top of a step they would be an LDS round trip in everybody's way
    int w, me, tm, pmo;      // position in the frame period, frame, ring time, (w mod Pt) (Q - 1)
    int lane;
    const Env<real, C> &e;

    BAND_FN Lane(const Env<real, C> &e_, int lane_, int slot_index) : e(e_) {
        lane = lane_;
        C z; z.x = 0; z.y = 0;
#pragma unroll
        for (int d = 0; d < NA; ++d) acc[d] = P{0, 0};
#pragma unroll
        for (int k = 0; k < (HELP ? 1 : LT + 1); ++k) { cn[k] = z; co[k] = z; }
        yE = 0;
        // a lane that has not started yet (its frame-time is negative) counts up to position 0 of its first frame
        w = -e.g.SKW * lane;
        me = lane;
        pmo = 0;
        // ring rows are addressed by the workgroup's time: a slot's main wave starts LAG steps after the slot before it (two steps
        // later still when helpers run a step ahead of it)
        tm = (int)(((long)e.g.LAG * slot_index + (HELP ? 1 : (NHELP ? 2 : 0))) % e.g.R);
#pragma unroll
        for (int k = 0; k <= LT; ++k) wfirst[k] = e.wt[(RLO + 1) * K1 + k];
        if constexpr (!HELP) {
#pragma unroll
            for (int k = 0; k <= LT; ++k) wown[k] = e.wt[k];
        }
    }
    // ring row written `age` steps before time t_mod.  A read issued for the NEXT step (t_mod = its time) sees rows of age >= 2: the
    // row of age 1 is being written while it is issued; a read of THIS step sees ages >= 1.
    BAND_FN int row_at(int t_mod, int age, int min_age) const {
        int x = t_mod - age;
#if !defined(__HIPCC__)
        assert(age >= min_age && age <= e.g.R - 2 + min_age);
#endif
        (void)min_age;
        return x < 0 ? x + e.g.R : x;
    }
    // the inputs of frame offset r + 1 at ring time tmx (pmx: the position's row of the twiddle table)
    BAND_FN In fetch(int r, int tmx, int pmx, int min_age) const {
        const Geom &g = e.g;
        In v;
        const int offL = (lane - (r + 1)) & (g.nls - 1);
        const int ageL = g.SKW * (r + 1) - LT - LEAD + (lane < r + 1 ? g.gap : 0);
        v.L = e.ring_own[(row_at(tmx, ageL, min_age) << g.lg) + offL];
        v.T = e.tw[pmx + r];
        if constexpr (!FIRST) {
            const int offR = (lane + r + 1) & (g.nls - 1);
            const int ageR = g.LAG - LT - LEAD - g.SKW * (r + 1) - (lane + r + 1 >= g.nls ? g.gap : 0);
            v.R = e.ring_prev[(row_at(tmx, ageR, min_age) << g.lg) + offR];
        } else {
            v.R = v.L;   // (replaced by the value from the skewed state)
        }
        return v;
    }
    BAND_FN void issue_lds(int tmx, int wx, int pmx) {
        const Geom &g = e.g;
        nx = fetch(RLO, tmx, pmx, 2);
        if constexpr (!HELP) {
            if constexpr (!FIRST) nxO = e.ring_prev[(row_at(tmx, g.LAG - LT, 2) << g.lg) + lane];
            int cx = wx - LT;
            if (cx < 0) cx += g.P;                        // still the images of the frame the lane has just left
            const int jj = cx - (g.F - 1);
            nxI = e.ring_own[(row_at(tmx, (jj >= 1 && jj <= LT) ? 2 * jj : 2, 2) << g.lg) + lane];
        }
    }
    BAND_FN void issue_global(int ux, int b) {
        const Geom &g = e.g;
        if constexpr (!HELP) pfA[b] = e.A[((long)ux << g.lg) + lane];
        if constexpr (FIRST) {
            const int NR = EXACT ? NRT : g.Q - 1;
            const C *Gu = e.G + ((long)(ux + LT) << g.lg);
            if constexpr (!HELP) pfO[b] = Gu[lane];
#pragma unroll
            for (int r = RLO; r < RHI; ++r) {
                if (r >= NR) continue;
                const int offR = (lane + r + 1) & (g.nls - 1);
                const int wrap = lane + r + 1 >= g.nls ? g.gap : 0;
                pfR[b][r - RLO] = Gu[((long)(g.SKW * (r + 1) + wrap) << g.lg) + offR];
            }
        }
    }
    BAND_FN void prologue() {   // before the step of frame-time 0
#pragma unroll
        for (int b = 0; b < PFD; ++b) issue_global(b, b);
        issue_lds(tm, w, pmo);
    }

    // the image below DC of position PH (= w): position -PH holds the conjugate -- of A' and B' too -- and reaches bins
    // ct = 0 .. LT - PH with V[r][ct + PH]; one lane at most (z), the others add zeros
    template <int PH> BAND_FN void images(const SD (&sd)[NB], bool z) {
        if constexpr (PH >= 1 && PH <= LT) {
            const int NR = EXACT ? NRT : e.g.Q - 1;
#pragma unroll
            for (int r = RLO; r < RHI; ++r) {
                if (r >= NR) continue;
                const C *wr = e.wt + (r + 1) * K1;
                // conj(A') + conj(B') = conj(S);  j (conj(A') - conj(B')) = (-D.x, D.y)
                const P Sc = vsel<real>(z, P{sd[r - RLO].S.x, -sd[r - RLO].S.y}, P{0, 0}), Dc = vsel<real>(z, P{-sd[r - RLO].D.x, sd[r - RLO].D.y}, P{0, 0});
#pragma unroll
                for (int ct = 0; ct <= LT - PH; ++ct) {
                    const C v = wr[ct + PH];
                    P a = acc[ct + LT - PH];
                    a = fma_x<real, C>(a, v, Sc);
                    a = fma_y<real, C>(a, v, Dc);
                    acc[ct + LT - PH] = a;
                }
            }
        }
    }

    // One step: the lane completes bin c = w - LT of its frame (if that is a bin) from the position w it receives.
    //   PB: which of the PFD prefetch buffers holds this step's loads (u mod PFD);  ph = u mod SKW = w mod SKW in every lane
    template <int PB> BAND_FN void step(int u, int ph) {
        const Geom &g = e.g;
        const int NR = EXACT ? NRT : g.Q - 1, F = g.F;
        // ---- this step's inputs were requested earlier (LDS: the first frame offset's one step early, the others' while the
        //      offset before them is worked on; skewed state: PFD steps early)
        int w1 = w + 1, me1 = me, pmo1 = pmo + NR == g.Pt * NR ? 0 : pmo + NR;
        if (w1 == g.P) { w1 = 0; me1 += g.nls; }
        if (w1 == 0) pmo1 = 0;
        const int tm1 = tm + 1 == g.R ? 0 : tm + 1;
        // what the helper waves left for this step a step ago (requested now, used when the bin is completed)
        C mb[NHELP ? 2 * NHELP : 1];
        if constexpr (NHELP > 0) {
#pragma unroll
            for (int h = 0; h < NHELP; ++h) {
                const C *cell = e.mail + ((((PB * NHELP + h) << g.lg) + lane) << 1);
                mb[2 * h] = cell[0];
                mb[2 * h + 1] = cell[1];
            }
        }

        C img; img.x = 0; img.y = 0;
        real amp = 0;
        if constexpr (!HELP) { img = nxI; amp = pfA[PB]; }
        const int c = w - LT;
        const bool act = w >= 0 && me < g.nls * g.nblk;
        if (ph == 0) {   // (wave-uniform) a lane starts a frame with empty sums
            const bool first = w == 0;
#pragma unroll
            for (int d = 0; d < NA; ++d) acc[d] = vsel<real>(first, P{0, 0}, acc[d]);
        }
        // ---- (a) old value of the frame itself at bin c + LT; an image above Nyquist whose source this sweep has already
        //      rewritten is the conjugate of that new value
        if constexpr (!HELP) {
            C O;
            if constexpr (FIRST) O = pfO[PB]; else O = nxO;
            const int kk = 2 * c + LT - 2 * (F - 1);
            C o = O;
#pragma unroll
            for (int k = 1; k <= LT; ++k) o = sel(kk == k, cj(cn[k]), o);
            co[LT] = o;
        }
        // ---- (b) neighbour frames: position w of frames me -+ r, turned by tau_r^w, reaches bins c .. c + 2 LT
        real y0 = 0;
        SD sd[NB];
        In cur = nx;
        C wcur[K1];              // the weights of the frame offset at hand, requested like its inputs: while the one before is worked on
#pragma unroll
        for (int k = 0; k <= LT; ++k) wcur[k] = wfirst[k];
#pragma unroll
        for (int r = RLO; r < RHI; ++r) {
            if (r >= NR) continue;
            const In in = cur;
            C wr[K1];
#pragma unroll
            for (int k = 0; k <= LT; ++k) wr[k] = wcur[k];
            if (r + 1 < RHI && r + 1 < NR) {
                cur = fetch(r + 1, tm, pmo, 1);
#pragma unroll
                for (int k = 0; k <= LT; ++k) wcur[k] = e.wt[(r + 2) * K1 + k];
            }
            C Rv;
            if constexpr (FIRST) Rv = pfR[PB][r - RLO]; else Rv = in.R;
            const P uu = pr<real>(in.L) + pr<real>(Rv), vv = pr<real>(in.L) - pr<real>(Rv);
            // A' = tau A, B' = conj(tau) B:  A' + B' = tau.x (A + B) + tau.y j (A - B),  j (A' - B') = tau.x j (A - B) - tau.y (A + B)
            SD t;
            t.S = fma_yj<real, C>(mul_x<real, C>(in.T, uu), in.T, vv);
            t.D = fnma_y<real, C>(mul_xj<real, C>(in.T, vv), in.T, uu);
            sd[r - RLO] = t;
            // DC and Nyquist (below): the imaginary part of the k = 0 taps
            y0 = fma_<real>(wr[0].x, t.S.y, y0);
            y0 = fma_<real>(wr[0].y, t.D.y, y0);
            {
                P a = acc[LT];
                a = fma_x<real, C>(a, wr[0], t.S);
                acc[LT] = fma_y<real, C>(a, wr[0], t.D);
            }
#pragma unroll
            for (int k = 1; k <= LT; ++k) {
                const C v = wr[k];
                P a = acc[LT + k], b = acc[LT - k];
                a = fma_x<real, C>(a, v, t.S);        // the taps below bin w + k:  V A' + conj(V) B'
                b = fma_x<real, C>(b, v, t.S);        // the taps above bin w - k:  V B' + conj(V) A'
                acc[LT + k] = fma_y<real, C>(a, v, t.D);
                acc[LT - k] = fnma_y<real, C>(b, v, t.D);
            }
#if defined(__HIPCC__)
            // (the requests of the next frame offset are in flight behind this one's arithmetic; moving more across this line only
            //  costs registers)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // images below DC: position -w is the conjugate of position w and reaches bins 0 .. LT - w (w = ph in the one lane that has
        // w <= LT; every lane has w = ph modulo SKW)
        if (ph >= 1 && ph <= LT) {
            const bool z = w == ph;
            switch (ph) {
                case 1: images<1>(sd, z); break;
                case 2: images<2>(sd, z); break;
                case 3: images<3>(sd, z); break;
                case 4: images<4>(sd, z); break;
                case 5: images<5>(sd, z); break;
                case 6: images<6>(sd, z); break;
                case 7: images<7>(sd, z); break;
                case 8: images<8>(sd, z); break;
                case 9: images<9>(sd, z); break;
                default: images<10>(sd, z); break;
            }
        }
        issue_lds(tm1, w1, pmo1);
        issue_global(u + PFD, PB);
        if constexpr (HELP) {
            // ---- a helper wave: what bin c gets from this wave's frame offsets, and their share of the DC / Nyquist imaginary part
            //      (position w), for the main wave's step of the same frame-time -- the next step of the workgroup
            C m0, m1;
            m0.x = acc[0].x; m0.y = acc[0].y;
            m1.x = y0; m1.y = 0;
            C *out = e.mail + ((((PB * e.mail_nh + e.mail_h) << g.lg) + lane) << 1);
            out[0] = m0;
            out[1] = m1;
#pragma unroll
            for (int d = 0; d < NA - 1; ++d) acc[d] = acc[d + 1];
            acc[NA - 1] = P{0, 0};
            w = w1; me = me1; tm = tm1; pmo = pmo1;
            return;
        } else {
        if constexpr (NHELP > 0) {
#pragma unroll
            for (int h = 0; h < NHELP; ++h) y0 += mb[2 * h + 1].x;
        }
        // DC and Nyquist.  Their neighbourhood is Hermitian (the images are exact conjugates), so in the reference every tap pair
        // k >= 1 adds x and then -x to the imaginary part of the sum, bit for bit (lwslib.cpp:310-311 on c = conj(b)): what is left
        // is the k = 0 taps -- exactly zero for a spectrogram whose DC / Nyquist bins are real, which they then stay.  That line
        // is unstable (lws_sys64.hip), and in scatter form the two halves of a pair arrive steps apart and cancel to
        // rounding only: so the imaginary part of these two bins is taken from the k = 0 taps alone, captured when their position
        // arrives (LT steps before the bin is complete).
        yE = (w == 0 || w == F - 1) ? y0 : yE;
        // ---- (c) the frame's own taps: new values below (the images below DC among them), old values above; k = 1 last (it is
        //      the value the previous step produced)
        if (ph == LT % g.SKW) {   // bin 0: nothing of this frame is new yet, the images below DC are those of the old values
            const bool b0 = c == 0;
#pragma unroll
            for (int k = 1; k <= LT; ++k) cn[k] = sel(b0, cj(co[k]), cn[k]);
        }
        P a0 = acc[0];
        if constexpr (NHELP > 0) {
#pragma unroll
            for (int h = 0; h < NHELP; ++h) a0 = a0 + pr<real>(mb[2 * h]);
        }
#pragma unroll
        for (int k = LT; k >= 1; --k) {
            const C wv = wown[k];
            const P b = pr<real>(cn[k]), cv = pr<real>(co[k]);
            a0 = fma_x<real, C>(a0, wv, b + cv);
            a0 = fma_yj<real, C>(a0, wv, b - cv);
        }
        // ---- re-projection on the target magnitude (lwslib.cpp:356-360)
        a0.y = (c == 0 || c == F - 1) ? yE : a0.y;
        const real m2 = a0.x * a0.x + a0.y * a0.y;
        const bool upd = act && c >= 0 && c <= F - 1 && me >= g.Q - 1 && me < g.T + g.Q - 1 && amp > e.thr && m2 > (real)0;
        const real sc = amp * rsqrt_(m2);
        C val;
        val.x = upd ? a0.x * sc : co[0].x;
        val.y = upd ? a0.y * sc : co[0].y;
        // ---- images above Nyquist (lwslib.cpp:365-367): written when the lane passes them, from its own ring
        {
            const bool before = c < 0;                                     // still the frame the lane has just left
            const int jj = (before ? c + g.P : c) - (F - 1);
            const bool has_frame = (before & (me >= g.nls)) | (!before & (me < g.nls * g.nblk));
            const bool is_img = (w >= 0) & has_frame & ((unsigned)(jj - 1) < (unsigned)LT);
            val = sel(is_img, cj(img), val);
            // an image above Nyquist that sits in the window of old values and whose source is the bin just written
            const int j2 = (F - 1) - c;
#pragma unroll
            for (int j = 1; 2 * j <= LT; ++j) co[2 * j] = sel(act && j2 == j, cj(val), co[2 * j]);
        }
        // (no position of a frame here -- before its bin 0, past its last image, before the first / after the last frame: val is
        //  the old value of such a row, which is zero, so zero is what gets written and the rows that are nobody's stay zero)
        e.ring_own[(tm << g.lg) + lane] = val;
        if (e.last) e.G[((long)u << g.lg) + lane] = val;
        // ---- windows move on by one bin
#pragma unroll
        for (int k = LT; k >= 2; --k) cn[k] = cn[k - 1];
        cn[1] = val;
        // ... and the image below DC of the bin just written, where the window of new values holds it: bin c = j is bin -j for
        // the step of bin c + 1, k = 2 j + 1 below it
#pragma unroll
        for (int j = 1; 2 * j + 1 <= LT; ++j) cn[2 * j + 1] = sel(c == j, cj(val), cn[2 * j + 1]);
#pragma unroll
        for (int k = 0; k < LT; ++k) co[k] = co[k + 1];
#pragma unroll
        for (int d = 0; d < NA - 1; ++d) acc[d] = acc[d + 1];
        acc[NA - 1] = P{0, 0};
        w = w1; me = me1; tm = tm1; pmo = pmo1;
        }
    }
};

}  // namespace band
}  // namespace lws
