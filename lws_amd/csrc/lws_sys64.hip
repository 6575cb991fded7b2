// lws_sys64.hip -- the batch sweeps of an fp64 plan (the reference's own arithmetic type, lwslib.h:6-26) as a systolic kernel.
//
// What it computes: LWSQ2 / LWSQ4 / LWSanyQ (lwslib.cpp:72-373) -- every bin of every frame in the reference's order,
// overwritten in place as soon as it is computed, Hermitian images kept in step (lwslib.cpp:356-368) -- for Q in {2, 4},
// L = 5, frames of up to ~2090 bins.  The sum of a bin is taken in a different ORDER than lwslib.cpp takes it (below), so
// results agree with the reference to rounding (1e-13 relative after 100 sweeps, tests/test_gpu_sys64.py), not bit for bit;
// the order-exact fp64 engine remains lws_generic.hip (LWS_FORCE_GENERIC).
//
// Design (tools/sys64_model.py is the same schedule in numpy, lane for lane, and was written first):
//   * one workgroup per spectrogram, one wave per sweep slot, lane = frame: 64 consecutive frames are in flight in a slot,
//     8 steps apart (> L + 1, and a multiple of Q so that `bin mod Q` -- the weight row -- is the same in every lane);
//     a lane updates one bin per step and takes up its next frame (64 further on) after P >= F + L steps.
//   * a slot's output lives in an LDS ring of R rows x 64 lanes x 16 B, one row per step (row = time mod R): the frames
//     m-1..m-(Q-1) of the same sweep are the lanes to the left a few rows back, the frames m+1..m+(Q-1) of the previous
//     sweep are in the ring of the slot before, LAG steps behind.  Slot 0 reads the previous pass from a time-skewed
//     copy of the state in HBM (coalesced rows, prefetched four steps ahead), the last slot writes it back in place.
//   * the taps of the neighbour frames are taken in SCATTER form: the position a step receives (one value per neighbour
//     frame) is combined once -- sum and difference of the two frames r apart, which is how lwslib.cpp:310-311 groups
//     them -- and added to the 2L+1 bins of the lane's frame it reaches, with 2 FMAs per component.  A bin is complete
//     when the position L bins above it has arrived; its own frame's taps come from two register windows (new values
//     below, old values above).  Per bin: 4(Q-1)(2L+1) + 8L FMAs instead of the 8-instruction pair of the gather form.
//   * Hermitian images: the L images above Nyquist are ordinary positions (a lane writes them as it passes, during the first
//     steps of its next frame); the images below DC are never stored -- the step that receives position w <= L also
//     scatters its conjugate as position -w.
//   * a step contains no branch: a wave is alone on its SIMD (the rings fill the LDS), and every s_cbranch in the step cost it
//     3-4 % (lane predicates are bitwise, one-lane work runs on zeroed inputs in the other lanes, every slot stores to HBM).
//   * frames of up to ~300 bins: two spectrograms side by side in a wave, 32 lanes each (a.nls), chosen per call by steps per
//     spectrogram and sweep.
//   * frames of more than ~525 bins (round 5, WPS = 2): 128 frames in flight, a sweep slot is two waves side by side on a ring row of 128
//     lanes (a lane period of 1024 steps; on 64 lanes such a frame needs a ring as deep as its surplus over 512 steps, one slot at best).
//     Same step, same flow control, same layout with rows of 128; chosen per call like the 32-lane geometry.  From ~1070 bins: 256 frames in
//     flight, four waves per slot (WPS = 4: 4096-point frames; one slot for Q = 4).
//
// Entry: launch_sys64 (lws_sys64.h), called by lws_capi.hip:run_stage for MODE_BATCH of an fp64 plan.
#include "lws_sys64.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace lws {
namespace {

constexpr int SL = 5;        // stencil half-width in bins
constexpr int NLN = 64;      // lanes = frames in flight per slot
constexpr int SKW = 8;       // steps between consecutive frames
constexpr int MARG = 96;     // rows before / after the skewed state that prefetches may touch
constexpr int PFD = 4;       // steps a global load is issued ahead of its use (2: no faster)
constexpr int LDS_ROWS = 160;
// spectrograms that go through the skewed scratch at a time (config 2's frames: 7.6 GB for 1024); LWS_S64_CHUNK: for tests
inline int chunk_size() {
    const char *v = getenv("LWS_S64_CHUNK");
    const int c = (v && *v) ? atoi(v) : 1024;
    return c >= 2 ? c & ~1 : 1024;   // (even: two spectrograms may share a workgroup)
}
constexpr uint64_t MASK_Q4 = 0xfd7fc3, MASK_Q2 = 0x5c3, MASK_ALL = ~0ull;   // non-zero weights of the default (sqrt-Hann) windows   // ring rows of 1 KB that fit the LDS

struct Geom { int P, gap, LAG, R, nblk, U, nls, rw; long rows; };   // rw: lanes of a ring row = 64 x waves per sweep slot
// nls: lanes (= frames in flight) per spectrogram -- 64, or 32 with two spectrograms side by side in a wave (short frames: a lane
// period of 64 x 8 steps would be half empty)
inline Geom geom(int F, int T, int Q, int nls) {
    Geom g;
    g.nls = nls;
    g.rw = nls > 64 ? nls : 64;
    const int NLN = nls;   // (shadows the wave width in the formulas below)
    const int Tp = T + 2 * (Q - 1);
    // steps a lane spends on a frame: its F bins, L steps before them (positions arrive L bins ahead); the L images above
    // Nyquist are written during the first steps of the lane's NEXT frame, which have no bin of their own
    g.P = std::max(NLN * SKW, (F + SL + 7) / 8 * 8);
    g.gap = g.P - NLN * SKW;
    g.LAG = (SL + SKW * (Q - 1) + g.gap + 2 + 7) / 8 * 8;
    g.R = g.LAG - SL + 1;
    g.nblk = (Tp + NLN - 1) / NLN;
    g.U = SKW * (NLN - 1) + g.P * g.nblk + 8;   // (+ 8: the images of the last frames)
    // rows after the skewed state: slot 0's prefetch of step ux <= U - 1 + PFD reads row ux + MARG + L + SKW r + (gap in the lanes
    // whose right-hand neighbour wrapped into the next block) for r <= Q - 1 -- past a fixed margin once gap > 64 (frames of more
    // than ~570 bins).  The values are discarded; the rows must exist.
    const int tail = std::max(MARG, PFD + SL + SKW * (Q - 1) + g.gap + 8);
    g.rows = (long)g.U + MARG + tail;
    return g;
}

struct S64Args {
    double2 *G;            // [B][rows][64] time-skewed state: frame me, bin c at row 8 (me % 64) + P (me / 64) + c + L
    const double *A;       // [B][rows][64] target magnitudes, same addressing
    const double *thr;     // [B][n_thr]
    long g_stride;         // rows * row width (64, or 128 with two waves per sweep slot)
    int n_thr, thr0, ns;   // this pass: sweeps thr0 .. thr0 + ns - 1
    int F, T, P, gap, LAG, R, nblk, U;
    int nls, B;            // lanes per spectrogram (64 / 32: one / two spectrograms per workgroup), spectrograms of the call
};

// The weights of row 0, W[0][r][k] (entries the reference skips -- |w| <= 1e-12, lws.pyx:227-232 -- are zero here).  The other
// rows are quarter turns of it (create_weights, lws.pyx:160-181: W[p][r][k] = W[0][r][k] exp(2 pi j p r / Q)), which costs
// nothing at compile time: the row of a step's bins is static in the eight-times unrolled loop.  They travel in the kernel
// arguments, i.e. they are scalar loads from the kernarg segment that the compiler may keep in SGPRs.
template <int Q> struct BaseW {
    double2 c[SL];            // r = 0, k = 1..L
    double2 n[Q - 1][SL + 1];   // r = 1..Q-1, k = 0..L
};

// acc += W * (s, d) with W = j^N V (CONJ: V conjugated first):  acc.x += W.x sx - W.y dy,  acc.y += W.x sy + W.y dx
// -- the grouped form of lwslib.cpp:310-311 on the sum / difference of the two frames r apart.
template <int N, bool CONJ> __device__ __forceinline__ void sc_add(double2 &acc, const double2 v, double sx, double dy, double sy, double dx) {
    const double vx = v.x, vy = CONJ ? -v.y : v.y;
    const double wx = N == 0 ? vx : (N == 1 ? -vy : (N == 2 ? -vx : vy));
    const double wy = N == 0 ? vy : (N == 1 ? vx : (N == 2 ? -vy : -vx));
    acc.x = __builtin_fma(wx, sx, acc.x);
    acc.x = __builtin_fma(-wy, dy, acc.x);
    acc.y = __builtin_fma(wx, sy, acc.y);
    acc.y = __builtin_fma(wy, dx, acc.y);
}
template <int Q> __host__ __device__ constexpr int quarter_turns(int row, int r) {   // of exp(2 pi j row r / Q), Q in {2, 4}
    return ((((row % Q) + Q) % Q) * r % Q) * (4 / Q);
}

__device__ __forceinline__ double2 cj(double2 v) { v.y = -v.y; return v; }
__device__ __forceinline__ double2 sel(bool c, double2 a, double2 b) { double2 r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; return r; }

// LDS writes of this step complete, then everybody meets.  (Not __syncthreads(): that waits for the global prefetches too.)
// the order of the independent parts of a step is chosen in the source (the bin's dependent chain interleaved with the taps of
// the bins above); this keeps the scheduler from undoing it
#define S64_PIN() __builtin_amdgcn_sched_barrier(0)
#define S64_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// MASK: bit r (L + 1) + k set if W[0][r][k] may be non-zero -- the taps of the other weights are not compiled in (the reference
// skips them at run time, lwslib.cpp:302,321).  Default windows: 0xfd7fc3 (Q = 4), 0x5c3 (Q = 2); anything else runs on MASK = ~0.
template <int Q, bool FIRST, uint64_t MASK, int WPS> struct Wave {
    static constexpr int L = SL, NR = Q - 1, NA = 2 * SL + 1;
    static constexpr int RW = NLN * WPS;   // lanes of a ring row: a sweep slot is WPS waves side by side (lane = place in the row)
    // lane state
    double2 acc[NA];          // sums of bins c .. c + 2L
    double2 cn[L + 1];        // cn[k]: new value of bin c - k
    double2 co[L + 1];        // co[k]: old value of bin c + k
    double yE;                // DC / Nyquist: the imaginary part of the bin's sum (see step)
    double2 nxL[NR], nxR[NR], nxO, nxI;   // inputs of the next step (LDS)
    double2 pfO[PFD], pfR[PFD][NR];       // slot 0: inputs of the next PFD steps (HBM)
    double pfA[PFD];
    int offL[NR], ageL[NR], offR[NR], ageR[NR], goffR[NR];
    int w, me, tm;
    // wave constants
    const S64Args &a;
    const BaseW<Q> &bw;
    double2 *ring_own;
    const double2 *ring_prev;
    double2 *G;
    const double *A;
    int lane, s;
    double thrA, thrB;        // thresholds of the workgroup's first / second spectrogram
    bool last, hi;            // hi: this lane works on the second one

    __device__ __forceinline__ Wave(const S64Args &a_, const BaseW<Q> &bw_, double2 *ring, double2 *G_, const double *A_, int s_, int lane_)
        : a(a_), bw(bw_), G(G_), A(A_), lane(lane_), s(s_) {
        ring_own = ring + (size_t)s * a.R * RW;
        ring_prev = ring + (size_t)(s > 0 ? s - 1 : 0) * a.R * RW;
        last = s == a.ns - 1;
        const int ls = lane & (a.nls - 1), hb = lane - ls, spw = RW / a.nls;
        hi = hb != 0;
        {
            const int b0 = blockIdx.x * spw, b1 = b0 + spw - 1 < a.B ? b0 + spw - 1 : a.B - 1, ts = a.thr0 + (s < a.ns ? s : 0);
            thrA = a.thr[(size_t)b0 * a.n_thr + ts];
            thrB = a.thr[(size_t)b1 * a.n_thr + ts];
        }
#pragma unroll
        for (int r = 1; r <= NR; ++r) {
            offL[r - 1] = hb + ((ls - r) & (a.nls - 1));
            ageL[r - 1] = SKW * r - L + (ls < r ? a.gap : 0);
            offR[r - 1] = hb + ((ls + r) & (a.nls - 1));
            const int wrap = ls + r >= a.nls ? a.gap : 0;
            ageR[r - 1] = a.LAG - L - SKW * r - wrap;
            goffR[r - 1] = (L + SKW * r + wrap) * RW + offR[r - 1];
        }
        double2 z; z.x = 0; z.y = 0;
#pragma unroll
        for (int d = 0; d < NA; ++d) acc[d] = z;
#pragma unroll
        for (int k = 0; k <= L; ++k) { cn[k] = z; co[k] = z; }
        w = -SKW * ls;
        me = ls;
        yE = 0;
        tm = (a.LAG * s) % a.R;
    }

    __device__ __forceinline__ int slot(int t_mod, int age) const {   // ring row written `age` steps before time t_mod
        int x = t_mod - age;
        return x < 0 ? x + a.R : x;
    }
    // LDS inputs of the step at ring time tmx, frame-time ux, lane position wx
    __device__ __forceinline__ void issue_lds(int tmx, int wx) {
#pragma unroll
        for (int r = 0; r < NR; ++r) nxL[r] = ring_own[slot(tmx, ageL[r]) * RW + offL[r]];
        if constexpr (!FIRST) {
            nxO = ring_prev[slot(tmx, a.LAG - L) * RW + lane];
#pragma unroll
            for (int r = 0; r < NR; ++r) nxR[r] = ring_prev[slot(tmx, ageR[r]) * RW + offR[r]];
        }
        int cx = wx - L;
        if (cx < 0) cx += a.P;                        // still the images of the frame the lane has just left
        const int jj = cx - (a.F - 1);
        nxI = ring_own[slot(tmx, (jj >= 1 && jj <= L) ? 2 * jj : 2) * RW + lane];
    }
    __device__ __forceinline__ void issue_global(int ux, int b) {
        const double *Au = A + (size_t)(ux + MARG) * RW;       // wave-uniform row pointers, per-lane constant offsets
        pfA[b] = Au[lane];
        if constexpr (FIRST) {
            const double2 *Gu = G + (size_t)(ux + MARG) * RW;
            pfO[b] = Gu[L * RW + lane];
#pragma unroll
            for (int r = 0; r < NR; ++r) pfR[b][r] = Gu[goffR[r]];
        }
    }
    __device__ __forceinline__ void prologue() {   // before the step of frame-time 0
#pragma unroll
        for (int b = 0; b < PFD; ++b) issue_global(b, b);
        issue_lds(tm, w);
    }

    // position w of the frames RR apart -> bins c + D .. c + 2L (target bin (PH - L + D) mod Q; weight W[row][RR][|D - L|] for
    // the taps below a bin, conj(W[-row][RR][|D - L|]) = j^(row RR) conj(W[0][RR][..]) for the taps above it)
    template <int PH, int RR, int D, int DEND = 2 * SL + 1> __device__ __forceinline__ void scatter(double sx, double dy, double sy, double dx) {
        if constexpr (D < DEND) {
            constexpr int K = D < L ? L - D : D - L;
            if constexpr ((MASK >> (RR * (L + 1) + K)) & 1)
                sc_add<quarter_turns<Q>(PH - L + D, RR), (D < L)>(acc[D], bw.n[RR - 1][K], sx, dy, sy, dx);
            scatter<PH, RR, D + 1, DEND>(sx, dy, sy, dx);
        }
    }
    // the image below DC of position PH (= w): position -PH, the conjugate, reaches bins CT = 0 .. L - PH with W[CT][RR][CT + PH]
    template <int PH, int RR, int CT> __device__ __forceinline__ void images(double sx, double dy, double sy, double dx) {
        if constexpr (CT <= L - PH) {
            constexpr int D = CT + L - PH, K = CT + PH;
            if constexpr ((MASK >> (RR * (L + 1) + K)) & 1)
                sc_add<quarter_turns<Q>(CT, RR), false>(acc[D], bw.n[RR - 1][K], sx, -dy, -sy, dx);
            images<PH, RR, CT + 1>(sx, dy, sy, dx);
        }
    }
    // (b) of a step for the frames RR, RR + 1, .. apart
    // (b) of a step, in two parts.  sd[r] = sum / difference of the two frames r + 1 apart at position w.  (Rows that are no
    // position of a frame -- before its bin 0, after its last image, before the first and after the last frame -- hold zeros:
    // in the ring because that is what a lane writes there, in HBM because the layout is cleared first; so a step that
    // receives no position adds nothing.)
    struct SD { double sx, dy, sy, dx; };
    // part 1: what the bin of this step still lacks (target D = 0), and the images below DC
    template <int PH, int RR> __device__ __forceinline__ void neighbours_now(const SD (&sd)[NR]) {
        if constexpr (RR <= NR) {
            const SD t = sd[RR - 1];
            scatter<PH, RR, 0, 1>(t.sx, t.dy, t.sy, t.dx);
            if constexpr (PH >= 1 && PH <= L) {
                // position -w is the conjugate of position w and reaches bins 0 .. L - w (one lane at most; the others add
                // zeros -- a branch around these few FMAs costs a wave that is alone on its SIMD more than they do)
                const bool z = w == PH;
                images<PH, RR, 0>(z ? t.sx : 0.0, z ? t.dy : 0.0, z ? t.sy : 0.0, z ? t.dx : 0.0);
            }
            neighbours_now<PH, RR + 1>(sd);
        }
    }
    // part 2: the bins above (targets 1 .. 2L) -- independent of this step's re-projection, which is one long dependent chain;
    // the caller places these between the links of that chain
    template <int PH, int RR, int REND> __device__ __forceinline__ void neighbours_later(const SD (&sd)[NR]) {
        if constexpr (RR <= REND && RR <= NR) {
            const SD t = sd[RR - 1];
            scatter<PH, RR, 1>(t.sx, t.dy, t.sy, t.dx);
            neighbours_later<PH, RR + 1, REND>(sd);
        }
    }

    template <int PH> __device__ __forceinline__ void step(int u) {
        // ---- this step's inputs (loaded earlier), then the loads of later steps
        double2 O, Rv[NR], Lv[NR];
        const double2 img = nxI;
        const double amp = pfA[PH % PFD];
#pragma unroll
        for (int r = 0; r < NR; ++r) Lv[r] = nxL[r];
        if constexpr (FIRST) {
            O = pfO[PH % PFD];
#pragma unroll
            for (int r = 0; r < NR; ++r) Rv[r] = pfR[PH % PFD][r];
        } else {
            O = nxO;
#pragma unroll
            for (int r = 0; r < NR; ++r) Rv[r] = nxR[r];
        }
        int w1 = w + 1, me1 = me;
        if (w1 == a.P) { w1 = 0; me1 += a.nls; }
        const int tm1 = tm + 1 == a.R ? 0 : tm + 1;
        issue_lds(tm1, w1);
        issue_global(u + PFD, PH % PFD);

        const int c = w - L, F = a.F;
        const bool act = w >= 0 && me < a.nls * a.nblk;
        if constexpr (PH == 0) {   // a lane starts a frame with empty sums
            const bool first = w == 0;
            double2 z; z.x = 0; z.y = 0;
#pragma unroll
            for (int d = 0; d < NA; ++d) acc[d] = sel(first, z, acc[d]);
        }
        // ---- (a) old value of the frame itself at bin c + L; an image above Nyquist whose source this sweep has already
        //      rewritten is the conjugate of that new value
        {
            const int kk = 2 * c + L - 2 * (F - 1);
            double2 o = O;
            o = sel(kk == 1, cj(cn[1]), o);
            o = sel(kk == 3, cj(cn[3]), o);
            o = sel(kk == 5, cj(cn[5]), o);
            co[L] = o;
        }
        // ---- (b) neighbour frames: position w of frames me -+ r reaches bins c .. c + 2L
        SD sd[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            sd[r].sx = Lv[r].x + Rv[r].x; sd[r].dy = Lv[r].y - Rv[r].y;
            sd[r].sy = Lv[r].y + Rv[r].y; sd[r].dx = Lv[r].x - Rv[r].x;
        }
        // DC and Nyquist.  Their neighbourhood is Hermitian (the images are exact conjugates), so in the reference every tap pair
        // k >= 1 adds x and then -x to the imaginary part of the sum, bit for bit (lwslib.cpp:310-311 on c = conj(b)): what is left
        // is the k = 0 taps -- exactly zero for a spectrogram whose DC / Nyquist bins are real, which they then stay.  That line
        // is unstable: an imaginary part of 1e-17 grows by three orders of magnitude per sweep (zero-phase input: a DC bin 4e-8
        // off the real axis at its first update, O(1) two sweeps later, 5 % of all bins after 100 sweeps).  In scatter form the
        // two halves of a pair arrive steps apart and cancel to rounding only, so the imaginary part of these two bins is taken
        // from the k = 0 taps alone, captured when their position arrives.  
        // (Position 0 arrives in phase 0, position F - 1 -- F is odd -- in an even phase, and their bins are complete L steps later:
        // the capture and its use are compiled into those phases only.  A half-length F - 1 that is 2 mod 4 puts the Nyquist bin
        // on weight row 2 when Q = 4: W[2][r][0] = (-1)^r W[0][r][0] -- exactly, here; in the reference's tensor numpy's
        // exp(j pi r) leaves those weights an imaginary part of 1e-16, which seeds the unstable line: for such frames the
        // reference's own Nyquist bins leave the real axis and its trajectory from a zero-phase start is that noise, amplified.)
        if constexpr (PH % 2 == 0) {
            constexpr bool ALT = (PH % 4 == 2) && Q == 4;
            double y0 = 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if ((MASK >> ((r + 1) * (L + 1))) & 1) {
                    const double sg = (ALT && (r % 2 == 0)) ? -1.0 : 1.0;     // frame distance r + 1 odd
                    y0 = __builtin_fma(sg * bw.n[r][0].x, sd[r].sy, y0);
                    y0 = __builtin_fma(sg * bw.n[r][0].y, sd[r].dx, y0);
                }
            yE = (((PH == 0) & (w == 0)) | (w == F - 1)) ? y0 : yE;
        }
        neighbours_now<PH, 1>(sd);
        // ---- (c) the frame's own taps: new values below (or their images below DC), old values above; k = 1 last (it is
        //      the value the previous step produced)
        double2 a0 = acc[0];
        constexpr int CC = ((PH - L) % 8 + 8) % 8;   // the bin below L a lane can be at in this phase
#pragma unroll
        for (int k = L; k >= 1; --k) {
            if (!((MASK >> k) & 1)) continue;
            double2 b = cn[k];
            if constexpr (CC < L) {
                if (k > CC) {
                    const int q = k - CC;
                    const double2 src = q < CC ? cn[CC - q > 0 ? CC - q : 0] : co[q - CC <= L ? q - CC : 0];
                    b = sel(c == CC, cj(src), b);
                }
            }
            const double2 wv = bw.c[k - 1];
            const double2 cv = co[k];
            a0.x = __builtin_fma(wv.x, b.x + cv.x, a0.x);
            a0.x = __builtin_fma(-wv.y, b.y - cv.y, a0.x);
            a0.y = __builtin_fma(wv.x, b.y + cv.y, a0.y);
            a0.y = __builtin_fma(wv.y, b.x - cv.x, a0.y);
        }
        // ---- re-projection on the target magnitude (lwslib.cpp:356-360)
        if constexpr ((PH - L) % 2 == 0) a0.y = ((c == 0) | (c == F - 1)) ? yE : a0.y;
        const double m2 = a0.x * a0.x + a0.y * a0.y;
        S64_PIN();
        neighbours_later<PH, 1, 1>(sd);
        S64_PIN();
        const bool upd = act && c >= 0 && c <= F - 1 && me >= Q - 1 && me < a.T + Q - 1 && amp > (hi ? thrB : thrA) && m2 > 0.0;
        const double sc = amp * rsqrt(m2);
        double2 val;
        val.x = upd ? a0.x * sc : co[0].x;
        val.y = upd ? a0.y * sc : co[0].y;
        // ---- images above Nyquist (lwslib.cpp:365-367): written when the lane passes them, from its own ring
        bool is_img;
        {
            // (bitwise on purpose: as && / ?: this became three branches in the middle of the step)
            const bool before = c < 0;                                     // still the frame the lane has just left
            const int jj = (before ? c + a.P : c) - (F - 1);
            const bool has_frame = (before & (me >= a.nls)) | (!before & (me < a.nls * a.nblk));
            is_img = (w >= 0) & has_frame & ((unsigned)(jj - 1) < (unsigned)L);
            val = sel(is_img, cj(img), val);
            const int j2 = (F - 1) - c;
            co[2] = sel(act && j2 == 1, cj(val), co[2]);
            co[4] = sel(act && j2 == 2, cj(val), co[4]);
        }
        // (no position of a frame here -- before its bin 0, past its last image, before the first / after the last frame: val is
        //  the old value of such a row, which is zero, so zero is what gets written and the invariant of `neighbours` holds)
        ring_own[tm * RW + lane] = val;
        // (every slot stores: the last one the row of the state, the others a row of the margin nobody reads -- no branch)
        G[(size_t)(last ? u + MARG : s) * RW + lane] = val;
        S64_PIN();
        neighbours_later<PH, 2, NR>(sd);
        // ---- windows move on by one bin
#pragma unroll
        for (int k = L; k >= 2; --k) cn[k] = cn[k - 1];
        cn[1] = val;
#pragma unroll
        for (int k = 0; k < L; ++k) co[k] = co[k + 1];
#pragma unroll
        for (int d = 0; d < NA - 1; ++d) acc[d] = acc[d + 1];
        acc[NA - 1].x = 0; acc[NA - 1].y = 0;
        w = w1; me = me1; tm = tm1;
    }
};

template <int Q, bool FIRST, uint64_t MASK, int WPS>
__device__ __forceinline__ void s64_wave(const S64Args &a, const BaseW<Q> &bw, double2 *ring, double2 *G, const double *A, int s, int lane) {
    Wave<Q, FIRST, MASK, WPS> wv(a, bw, ring, G, A, s, lane);
    const bool live = s < a.ns;
    const int t_end = a.U + a.LAG * (a.ns - 1);   // U and LAG are multiples of 8
    for (int t0 = 0; t0 < t_end; t0 += 8) {
        const int u0 = t0 - a.LAG * s;
        if (live && u0 >= 0 && u0 < a.U) {
            if (u0 == 0) wv.prologue();
            wv.template step<0>(u0); S64_BARRIER();
            wv.template step<1>(u0 + 1); S64_BARRIER();
            wv.template step<2>(u0 + 2); S64_BARRIER();
            wv.template step<3>(u0 + 3); S64_BARRIER();
            wv.template step<4>(u0 + 4); S64_BARRIER();
            wv.template step<5>(u0 + 5); S64_BARRIER();
            wv.template step<6>(u0 + 6); S64_BARRIER();
            wv.template step<7>(u0 + 7); S64_BARRIER();
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) S64_BARRIER();
        }
    }
}

template <int Q, int NS, uint64_t MASK, int WPS>
__global__ void __launch_bounds__(NLN * NS * WPS) k_sys64(S64Args a, BaseW<Q> bw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s64_lds[];
    double2 *ring = reinterpret_cast<double2 *>(s64_lds);
    constexpr int RW = NLN * WPS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = wave / WPS;                                      // sweep slot
    const int lane = (wave % WPS) * NLN + (threadIdx.x & (NLN - 1));   // place in the slot's ring row (WPS = 2: hardware waves 2s and 2s + 1 hold its two halves)
    double2 *G = a.G + (size_t)blockIdx.x * a.g_stride;
    const double *A = a.A + (size_t)blockIdx.x * a.g_stride;
    {
        double2 z; z.x = 0; z.y = 0;
        for (int i = threadIdx.x; i < NS * a.R * RW; i += NLN * NS * WPS) ring[i] = z;
        __syncthreads();
    }
    if (s == 0) s64_wave<Q, true, MASK, WPS>(a, bw, ring, G, A, s, lane);
    else s64_wave<Q, false, MASK, WPS>(a, bw, ring, G, A, s, lane);
}

// extended buffers [B][Tp][Np] <-> the skewed layout
__global__ void k_s64_load(const double2 *state, const double *amp, double2 *G, double *A, int F, int Tp, int P, long g_stride, int nls, int rw) {
    const int bb = blockIdx.y, me = blockIdx.x, Np = F + 2 * SL;
    const int spw = rw / nls, b = bb / spw, ls = me & (nls - 1), j = (bb % spw) * nls + ls, blk = me / nls;   // b: workgroup
    const long base = (long)SKW * ls + (long)P * blk + SL + MARG;
    const double2 *src = state + ((size_t)bb * Tp + me) * Np + SL;
    const double *asrc = amp + ((size_t)bb * Tp + me) * Np + SL;
    for (int c = threadIdx.x; c < F + SL; c += blockDim.x) {
        G[(size_t)b * g_stride + (base + c) * rw + j] = src[c];
        A[(size_t)b * g_stride + (base + c) * rw + j] = asrc[c];
    }
}
__global__ void k_s64_store(double2 *state, const double2 *G, int F, int Tp, int P, long g_stride, int nls, int rw) {
    const int bb = blockIdx.y, me = blockIdx.x, Np = F + 2 * SL;
    const int spw = rw / nls, b = bb / spw, ls = me & (nls - 1), j = (bb % spw) * nls + ls, blk = me / nls;
    const long base = (long)SKW * ls + (long)P * blk + SL + MARG;
    double2 *dst = state + ((size_t)bb * Tp + me) * Np + SL;
    for (int c = threadIdx.x; c < F + SL; c += blockDim.x) {
        const double2 v = G[(size_t)b * g_stride + (base + c) * rw + j];
        dst[c] = v;
        if (c >= 1 && c <= SL) dst[-c] = cj(v);
    }
}

template <int Q, int NS, uint64_t MASK, int WPS = 1>
hipError_t launch_pass(const S64Args &a, const BaseW<Q> &bw, int B, hipStream_t stream) {
    static std::atomic<unsigned long long> done{0};
    constexpr int RW = NLN * WPS;
    const size_t lds = (size_t)NS * a.R * RW * sizeof(double2);
    int dev = 0;
    if (attr_needed(done, &dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sys64<Q, NS, MASK, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done(done, dev);
    }
    k_sys64<Q, NS, MASK, WPS><<<dim3((B + RW / a.nls - 1) / (RW / a.nls)), dim3(NLN * NS * WPS), lds, stream>>>(a, bw);
    return hipGetLastError();
}

int slots_for(int Q, int R, int rw) {   // sweep slots per workgroup: what the LDS holds, among the builds that exist
    const int fit = (LDS_ROWS * NLN / rw) / R;
    if (rw > 2 * NLN) {   // four waves per slot (frames of ~1070 to ~2090 bins)
        if (Q == 4) return fit >= 1 ? 1 : 0;
        if (Q == 2) return fit >= 2 ? 2 : (fit >= 1 ? 1 : 0);
        return 0;
    }
    if (rw > NLN) {   // two waves per slot (frames of ~525 to ~1070 bins)
        if (Q == 4) return fit >= 2 ? 2 : (fit >= 1 ? 1 : 0);
        if (Q == 2) return fit >= 4 ? 4 : (fit >= 2 ? 2 : (fit >= 1 ? 1 : 0));
        return 0;
    }
    if (Q == 4) return fit >= 4 ? 4 : (fit >= 3 ? 3 : (fit >= 1 ? 1 : 0));   // (one wave per SIMD: a sweep slot uses > 256 registers)
    if (Q == 2) return fit >= 8 ? 8 : (fit >= 5 ? 5 : (fit >= 3 ? 3 : (fit >= 1 ? 1 : 0)));
    return 0;
}

// W (host, complex128 interleaved, [Qp][Q][L+1]): every row a quarter-turn image of row 0, to rounding?  Fills the row-0 weights.
template <int Q> bool base_weights(const double *W, int Qp, BaseW<Q> *out) {
    const int K1 = SL + 1;
    if (!W || Qp < 1 || Qp % Q != 0) return false;   // (Qp - row) mod Qp must be -row mod Q
    auto at = [&](int p, int r, int k, int c) { return W[2 * (((size_t)p * Q + r) * K1 + k) + c]; };
    double scale = 0;
    for (size_t x = 0; x < (size_t)Qp * Q * K1; ++x) scale = std::max(scale, std::hypot(W[2 * x], W[2 * x + 1]));
    if (!(scale > 0)) return false;
    for (int p = 0; p < Qp; ++p)
        for (int r = 0; r < Q; ++r)
            for (int k = 0; k < K1; ++k) {
                if (r == 0 && k == 0) continue;   // never read (update == 2)
                const double br = at(0, r, k, 0), bi = at(0, r, k, 1);
                const int n = quarter_turns<Q>(p, r);
                const double er = n == 0 ? br : (n == 1 ? -bi : (n == 2 ? -br : bi));
                const double ei = n == 0 ? bi : (n == 1 ? br : (n == 2 ? -bi : -br));
                if (std::hypot(at(p, r, k, 0) - er, at(p, r, k, 1) - ei) > 1e-13 * scale) return false;
                // the reference skips a weight by its own magnitude (lws.pyx:232); rows that disagree about that cannot share row 0
                if ((std::hypot(at(p, r, k, 0), at(p, r, k, 1)) > 1e-12) != (std::hypot(br, bi) > 1e-12)) return false;
            }
    if (out) {
        auto put = [&](double2 &d, int r, int k) {
            const bool on = std::hypot(at(0, r, k, 0), at(0, r, k, 1)) > 1e-12;
            d.x = on ? at(0, r, k, 0) : 0.0;
            d.y = on ? at(0, r, k, 1) : 0.0;
        };
        for (int k = 1; k <= SL; ++k) put(out->c[k - 1], 0, k);
        for (int r = 1; r < Q; ++r)
            for (int k = 0; k <= SL; ++k) put(out->n[r - 1][k], r, k);
    }
    return true;
}

// The geometry a call runs on: one spectrogram per workgroup, or two side by side when that takes fewer steps per spectrogram
// and sweep (frames of up to ~300 bins).  NS = 0: no ring fits.
Geom choose_geom(int F, int T, int Q, int *NS_out) {
    Geom best = geom(F, T, Q, NLN);
    int best_ns = slots_for(Q, best.R, best.rw);
    double best_cost = best_ns ? (double)(best.U + best.LAG * (best_ns - 1)) / best_ns : 1e300;
    const Geom h = geom(F, T, Q, NLN / 2);
    const int h_ns = slots_for(Q, h.R, h.rw);
    if (h_ns) {
        const double cost = (double)(h.U + h.LAG * (h_ns - 1)) / h_ns / 2;
        if (cost < best_cost) { best = h; best_ns = h_ns; best_cost = cost; }
    }
    // frames too long for a 64-lane period (a ring as deep as the surplus): 128 frames in flight, two waves per sweep slot
    const Geom w = geom(F, T, Q, 2 * NLN);
    const int w_ns = slots_for(Q, w.R, w.rw);
    if (w_ns) {
        const double cost = (double)(w.U + w.LAG * (w_ns - 1)) / w_ns;
        if (cost < best_cost) { best = w; best_ns = w_ns; best_cost = cost; }
    }
    // and 256 (four waves per slot: 4096-point frames)
    const Geom x = geom(F, T, Q, 4 * NLN);
    const int x_ns = slots_for(Q, x.R, x.rw);
    if (x_ns) {
        const double cost = (double)(x.U + x.LAG * (x_ns - 1)) / x_ns;
        if (cost < best_cost) { best = x; best_ns = x_ns; best_cost = cost; }
    }
    *NS_out = best_ns;
    return best;
}

}  // namespace

bool sys64_supports(int F, int T, int L, int Q, int Qp, int update, const double *W) {
    if (L != SL || update != 2 || T < 1 || F < 2 * SL + 7) return false;
    if (Q != 2 && Q != 4) return false;
    // even F (a frame length that is 2 mod 4): the Nyquist bin would fall into an odd phase of the unrolled step, where its
    // exact-real handling (yE) is not compiled in; the order-exact engine takes those plans
    if (!(F & 1)) return false;
    int ns = 0;
    (void)choose_geom(F, T, Q, &ns);
    if (ns < 1) return false;
    return Q == 4 ? base_weights<4>(W, Qp, nullptr) : base_weights<2>(W, Qp, nullptr);
}

size_t sys64_bytes(int B, int F, int T, int Q, size_t *amp_bytes) {
    int ns = 0;
    const Geom g = choose_geom(F, T, Q, &ns);
    const size_t wgs = (std::min(B, chunk_size()) + g.rw / g.nls - 1) / (g.rw / g.nls);
    if (amp_bytes) *amp_bytes = wgs * g.rows * g.rw * sizeof(double);
    return wgs * g.rows * g.rw * sizeof(double2);
}

bool sys64_layout(int F, int T, int Q, long out[4]) {
    int ns = 0;
    const Geom g = choose_geom(F, T, Q, &ns);
    if (ns < 1) return false;
    out[0] = g.rows;                                                            // rows of the skewed state / magnitudes per workgroup
    out[1] = (long)(g.U - 1 + PFD) + MARG + SL + SKW * (Q - 1) + g.gap;         // highest row slot 0's prefetch reads (issue_global)
    out[2] = (long)(g.U - 1) + MARG;                                            // highest row a step writes
    out[3] = g.gap;
    return true;
}

const char *sys64_name(int F, int T, int Q) {
    int ns = 0;
    const int rw = choose_geom(F, T, Q, &ns).rw;
    if (rw > 2 * NLN) return Q == 2 ? "systolic_fp64_q2_xwide" : "systolic_fp64_q4_xwide";
    return Q == 2 ? (rw > NLN ? "systolic_fp64_q2_wide" : "systolic_fp64_q2") : (rw > NLN ? "systolic_fp64_q4_wide" : "systolic_fp64_q4");
}

namespace {
template <int Q>
hipError_t run_passes(S64Args a, const double *W, int Qp, int NS, int wps, int n_thr, int B, hipStream_t stream, int *n_out) {
    BaseW<Q> bw;
    if (!base_weights<Q>(W, Qp, &bw)) return hipErrorInvalidValue;
    // the build without the taps that the default windows' weights do not have, if this tensor has none of them either
    uint64_t mask = 0;
    for (int k = 1; k <= SL; ++k) mask |= (uint64_t)(bw.c[k - 1].x != 0 || bw.c[k - 1].y != 0) << k;
    for (int r = 1; r < Q; ++r)
        for (int k = 0; k <= SL; ++k) mask |= (uint64_t)(bw.n[r - 1][k].x != 0 || bw.n[r - 1][k].y != 0) << (r * (SL + 1) + k);
    const bool dflt = (mask & ~(Q == 4 ? MASK_Q4 : MASK_Q2)) == 0;
    int n = 0;
    for (int i0 = 0; i0 < n_thr; i0 += NS, ++n) {
        a.thr0 = i0;
        a.ns = std::min(NS, n_thr - i0);
        hipError_t e;
        if (wps == 4) {
            if constexpr (Q == 4) e = dflt ? launch_pass<4, 1, MASK_Q4, 4>(a, bw, B, stream) : launch_pass<4, 1, MASK_ALL, 4>(a, bw, B, stream);
            else {
                if (dflt) e = NS == 2 ? launch_pass<2, 2, MASK_Q2, 4>(a, bw, B, stream) : launch_pass<2, 1, MASK_Q2, 4>(a, bw, B, stream);
                else e = NS == 2 ? launch_pass<2, 2, MASK_ALL, 4>(a, bw, B, stream) : launch_pass<2, 1, MASK_ALL, 4>(a, bw, B, stream);
            }
        } else if (wps == 2) {
            if constexpr (Q == 4) {
                if (dflt) e = NS == 2 ? launch_pass<4, 2, MASK_Q4, 2>(a, bw, B, stream) : launch_pass<4, 1, MASK_Q4, 2>(a, bw, B, stream);
                else e = NS == 2 ? launch_pass<4, 2, MASK_ALL, 2>(a, bw, B, stream) : launch_pass<4, 1, MASK_ALL, 2>(a, bw, B, stream);
            } else {
                if (dflt) e = NS == 4 ? launch_pass<2, 4, MASK_Q2, 2>(a, bw, B, stream) : (NS == 2 ? launch_pass<2, 2, MASK_Q2, 2>(a, bw, B, stream) : launch_pass<2, 1, MASK_Q2, 2>(a, bw, B, stream));
                else e = NS == 4 ? launch_pass<2, 4, MASK_ALL, 2>(a, bw, B, stream) : (NS == 2 ? launch_pass<2, 2, MASK_ALL, 2>(a, bw, B, stream) : launch_pass<2, 1, MASK_ALL, 2>(a, bw, B, stream));
            }
        } else if constexpr (Q == 4) {
            if (dflt) e = NS == 4 ? launch_pass<4, 4, MASK_Q4>(a, bw, B, stream) : (NS == 3 ? launch_pass<4, 3, MASK_Q4>(a, bw, B, stream) : launch_pass<4, 1, MASK_Q4>(a, bw, B, stream));
            else e = NS == 4 ? launch_pass<4, 4, MASK_ALL>(a, bw, B, stream) : (NS == 3 ? launch_pass<4, 3, MASK_ALL>(a, bw, B, stream) : launch_pass<4, 1, MASK_ALL>(a, bw, B, stream));
        } else {
            if (dflt) e = NS == 8 ? launch_pass<2, 8, MASK_Q2>(a, bw, B, stream) : (NS == 5 ? launch_pass<2, 5, MASK_Q2>(a, bw, B, stream) : (NS == 3 ? launch_pass<2, 3, MASK_Q2>(a, bw, B, stream) : launch_pass<2, 1, MASK_Q2>(a, bw, B, stream)));
            else e = NS == 8 ? launch_pass<2, 8, MASK_ALL>(a, bw, B, stream) : (NS == 5 ? launch_pass<2, 5, MASK_ALL>(a, bw, B, stream) : (NS == 3 ? launch_pass<2, 3, MASK_ALL>(a, bw, B, stream) : launch_pass<2, 1, MASK_ALL>(a, bw, B, stream)));
        }
        if (e != hipSuccess) return e;
    }
    *n_out = n;
    return hipSuccess;
}
}  // namespace

hipError_t launch_sys64(const GenericArgs<double> &ga, const double *W_host, int B, void *gs, void *gamp, hipStream_t stream, int *launches,
                        hipEvent_t ev0, hipEvent_t ev1) {
    if (B <= 0 || ga.n_thr <= 0) return hipSuccess;
    const int F = ga.F, T = ga.T, Q = ga.Q, Tp = T + 2 * (Q - 1);
    int NS = 0;
    const Geom g = choose_geom(F, T, Q, &NS);
    if (NS < 1 || ga.mode != MODE_BATCH || ga.L != SL) return hipErrorInvalidValue;
    double2 *G = static_cast<double2 *>(gs);
    double *A = static_cast<double *>(gamp);
    const long g_stride = g.rows * g.rw;
    const size_t Np = F + 2 * SL;
    hipError_t e;
    if (ev0) (void)hipEventRecord(ev0, stream);
    int n_all = 0;
    // a batch larger than the scratch was sized for (sys64_bytes: at most CHUNK spectrograms) goes through it chunk by chunk
    const int CHUNK = chunk_size();
    for (int b0 = 0; b0 < B; b0 += CHUNK) {
        const int Bc = std::min(CHUNK, B - b0);
        const size_t wgs = (Bc + g.rw / g.nls - 1) / (g.rw / g.nls);
        // rows no frame owns are read by lanes whose results are discarded; they must still be numbers the first time
        if ((e = hipMemsetAsync(G, 0, wgs * g_stride * sizeof(double2), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(A, 0, wgs * g_stride * sizeof(double), stream)) != hipSuccess) return e;
        double2 *state = ga.state + (size_t)b0 * Tp * Np;
        k_s64_load<<<dim3(Tp, Bc), 256, 0, stream>>>(state, ga.amp + (size_t)b0 * Tp * Np, G, A, F, Tp, g.P, g_stride, g.nls, g.rw);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        S64Args a;
        a.G = G; a.A = A; a.thr = ga.thr + (size_t)b0 * ga.n_thr; a.g_stride = g_stride;
        a.n_thr = ga.n_thr; a.thr0 = 0; a.ns = 0;
        a.F = F; a.T = T; a.P = g.P; a.gap = g.gap; a.LAG = g.LAG; a.R = g.R; a.nblk = g.nblk; a.U = g.U;
        a.nls = g.nls; a.B = Bc;
        int n = 0;
        e = Q == 4 ? run_passes<4>(a, W_host, ga.Qp, NS, g.rw / NLN, ga.n_thr, Bc, stream, &n) : run_passes<2>(a, W_host, ga.Qp, NS, g.rw / NLN, ga.n_thr, Bc, stream, &n);
        if (e != hipSuccess) return e;
        n_all += n;
        k_s64_store<<<dim3(Tp, Bc), 256, 0, stream>>>(state, G, F, Tp, g.P, g_stride, g.nls, g.rw);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (ev1) (void)hipEventRecord(ev1, stream);
    if (launches) *launches = n_all;
    return hipSuccess;
}

}  // namespace lws
