// lws_generic.hip -- order-exact "generic" LWS engine: works for every shape, weight tensor and
// mode the reference accepts (batch / no-future / bug-compatible no-future Q4 / online), in fp32 or
// fp64.  One workgroup owns one spectrogram; the extended spectrogram lives in global memory in
// the reference's own layout ([Tp][Np], physical Hermitian pad columns and edge-pad frames), and
// the in-place Gauss-Seidel sweep of lwslib.cpp is replayed as a skewed wavefront:
//
//     bin (frame m, bin c) of sweep s runs at step  t = D*s + (L+1)*m + c ,  D = Q*(L+1)
//
// All bins with equal t are independent: the neighbours a bin needs "new" -- (m, c-k),
// (m-r, c+-k) -- have strictly smaller t, the ones it needs "old" -- (m, c+k), (m+r, c+-k) --
// strictly larger t, and a later sweep is never less than D = L + (L+1)(Q-1) + 1 steps behind the
// previous one, which is the distance to the farthest neighbour (skew
// L+1 and lag 24/48 reproduce the sequential result to 5e-15, skew L / lag-1 do not).  The same
// argument holds for the online driver's sweeps over 1..LA+1 frames because D is applied between
// *every* pair of consecutive sweeps in the reference's call order.
//
// This kernel is the reference-semantics workhorse (and the fp64 parity anchor); the fast path for
// the headline batch workload is the systolic kernel in lws_systolic.hip.
#include "lws_common.h"

#include <type_traits>

namespace lws {

// a += w*b + conj(w)*c in the grouped form of lwslib.cpp:310-311: cancels exactly when c == conj(b)
// (Hermitian images / real input), which keeps ill-conditioned bins in step with the CPU arithmetic.
template <typename real, typename C>
__device__ __forceinline__ void pair(C &a, const C w, const C b, const C c) {
    a.x += w.x * (b.x + c.x) - w.y * (b.y - c.y);
    a.y += w.x * (b.y + c.y) + w.y * (b.x - c.x);
}
template <typename real, typename C>
__device__ __forceinline__ void mac(C &a, const C w, const C s) {  // a += w * s
    a.x += w.x * s.x - w.y * s.y;
    a.y += w.x * s.y + w.y * s.x;
}
template <typename real, typename C>
__device__ __forceinline__ void macc(C &a, const C w, const C s) {  // a += conj(w) * s
    a.x += w.x * s.x + w.y * s.y;
    a.y += w.x * s.y - w.y * s.x;
}

// The local weighted sum of one bin (no write): LWSanyQ (lwslib.cpp:297-354), NoFuture_LWSanyQ
// (634-671) and Asym_UpdatePhaseanyQ (1157-1253) in one body; `centre` / `two_sided` are the
// reference's cframe / rframe (lwslib.cpp:1143-1151).
// `mk` (may be null): the participation flags of each weight row packed into one 64-bit word (bit r*(L+1)+k), built once per
// workgroup in LDS.  The reference tests a flag per weight (lwslib.cpp:302,321,...); tested as bytes from memory, every tap
// waits for its flag before its loads can even be issued -- ~150 serialised round trips per bin; as bits of a register the
// tests are immediate and the tap loads of a bin are in flight together.  Same taps, same order, same arithmetic.
struct FlagView {
    const uint8_t *bytes;
    unsigned long long mask;
    bool packed;
    __device__ __forceinline__ bool operator[](int i) const { return packed ? ((mask >> i) & 1ull) != 0 : bytes[i] != 0; }
};
// Where the neighbours of a bin are.  RowView: the reference's extended layout [Tp][Np] (frame dr away = dr rows away).
// SkewView: the time-skewed layout of the batch engine below, SW[time][frame mod NL] with time = sk * frame + column:
// the neighbour (dr, dk) of a bin is  sk*dr + dk  rows and dr lanes away, so the taps of all the bins a wavefront step
// updates -- consecutive frames at the same time -- are consecutive elements of one row.
template <typename C> struct RowView {
    const C *ctr; int Np;
    __device__ __forceinline__ C at(int dr, int dk) const { return ctr[(long)dr * Np + dk]; }
};
template <typename C> struct SkewView {
    const C *sw; long row; int lane, sk, nl;   // nl: lanes of a row (power of two)
    __device__ __forceinline__ C at(int dr, int dk) const { return sw[(row + (long)sk * dr + dk) * nl + ((lane + dr) & (nl - 1))]; }
};
template <typename real, typename View>
__device__ __forceinline__ typename cx<real>::type accumulate_v(const View nb, int c, bool centre, int two_sided,
                                                                const WeightSet<real> ws, int L, int Q, int Qp, bool add_self,
                                                                real qdiv, const unsigned long long *mk) {
    using C = typename cx<real>::type;
    const int K1 = L + 1, RQ = Q * K1;
    const int row = c % Qp, rowneg = (Qp - row) % Qp;
    const C *wa = ws.w + (size_t)row * RQ, *wb = ws.w + (size_t)rowneg * RQ;
    const FlagView fa{ws.flag + (size_t)row * RQ, mk ? mk[row] : 0ull, mk != nullptr};
    const FlagView fb{ws.flag + (size_t)rowneg * RQ, mk ? mk[rowneg] : 0ull, mk != nullptr};
    C a;
    a.x = 0; a.y = 0;
    if (centre) {
        if (add_self) { const C s0 = nb.at(0, 0); a.x += s0.x / qdiv; a.y += s0.y / qdiv; }
        for (int k = 1; k <= L; ++k)
            if (fa[k]) pair<real>(a, wa[k], nb.at(0, -k), nb.at(0, k));
    }
    for (int r = 1; r < Q; ++r) {
        const int u = r * K1;
        const bool both = r < two_sided;
        if (fa[u]) {
            if (both) pair<real>(a, wa[u], nb.at(-r, 0), nb.at(r, 0));
            else mac<real>(a, wa[u], nb.at(-r, 0));
        }
        for (int k = 1; k <= L; ++k) {
            if (fa[u + k]) {
                if (both) pair<real>(a, wa[u + k], nb.at(-r, -k), nb.at(r, -k));
                else mac<real>(a, wa[u + k], nb.at(-r, -k));
            }
            if (fb[u + k]) {
                if (both) pair<real>(a, wb[u + k], nb.at(r, k), nb.at(-r, k));
                else macc<real>(a, wb[u + k], nb.at(-r, k));
            }
        }
    }
    return a;
}
template <typename real>
__device__ __forceinline__ typename cx<real>::type accumulate(const typename cx<real>::type *ctr,
                                                              int c, bool centre, int two_sided,
                                                              const WeightSet<real> ws, int Np,
                                                              int L, int Q, int Qp, bool add_self,
                                                              real qdiv, const unsigned long long *mk = nullptr) {
    using C = typename cx<real>::type;
    return accumulate_v<real>(RowView<C>{ctr, Np}, c, centre, two_sided, ws, L, Q, Qp, add_self, qdiv, mk);
}
// One bin update: weighted sum, magnitude re-projection (lwslib.cpp:356-360), Hermitian image upkeep.
template <typename real>
__device__ __forceinline__ void update_bin(typename cx<real>::type *S, const real *amp, int m_ext,
                                           int c, bool centre, int two_sided,
                                           const WeightSet<real> ws, real thr, int F, int L, int Q,
                                           int Qp, bool add_self, real qdiv, const unsigned long long *mk = nullptr) {
    using C = typename cx<real>::type;
    const int Np = F + 2 * L;
    const int n = c + L;
    const size_t idx = (size_t)m_ext * Np + n;
    const real target = amp[idx];
    if (!(target > thr)) return;
    C *ctr = S + idx;
    const C a = accumulate<real>(ctr, c, centre, two_sided, ws, Np, L, Q, Qp, add_self, qdiv, mk);
    const real mag = sqrt(a.x * a.x + a.y * a.y);
    if (!(mag > 0)) return;
    C v;
    v.x = a.x * target / mag;
    v.y = a.y * target / mag;
    ctr[0] = v;
    // keep the Hermitian images in the pad columns in sync (lwslib.cpp:362-367)
    const int nyq = F + L - 1;
    C vc;
    vc.x = v.x; vc.y = -v.y;
    if (n >= L + 1 && n < 2 * L + 1) S[(size_t)m_ext * Np + 2 * L - n] = vc;
    else if (n >= F - 1 && n < nyq) S[(size_t)m_ext * Np + 2 * nyq - n] = vc;
}

// One bin of NoFuture_LWSQ4 as shipped (lwslib.cpp:550-613): flat offset (m-r)*Np + 2n +- k.
template <typename real>
__device__ __forceinline__ void update_bin_nfq4(typename cx<real>::type *S, const real *amp,
                                                int m_ext, int c, const WeightSet<real> ws,
                                                real thr, int F, int L, const unsigned long long *mk = nullptr) {
    using C = typename cx<real>::type;
    const int Q = 4, Np = F + 2 * L, n = c + L, K1 = L + 1, RQ = Q * K1;
    const size_t idx = (size_t)m_ext * Np + n;
    const real target = amp[idx];
    if (!(target > thr)) return;
    const int row = c % Q;
    const C *wa = ws.w + (size_t)row * RQ;
    const FlagView fa{ws.flag + (size_t)row * RQ, mk ? mk[row] : 0ull, mk != nullptr};
    C a;
    a.x = 0; a.y = 0;
    for (int r = Q - 1; r > 0; --r) {
        const C *p = S + (size_t)(m_ext - r) * Np + 2 * (size_t)n;
        const int u = r * K1;
        const real sgn = ((c & 1) && (r & 1)) ? (real)-1 : (real)1;
        for (int k = 1; k <= L; ++k)
            if (fa[u + k]) {
                C hi = p[k];
                hi.x *= sgn; hi.y *= sgn;
                pair<real>(a, wa[u + k], p[-k], hi);
            }
        if (fa[u]) mac<real>(a, wa[u], p[0]);
    }
    const real mag = sqrt(a.x * a.x + a.y * a.y);
    if (!(mag > 0)) return;
    C v;
    v.x = a.x * target / mag;
    v.y = a.y * target / mag;
    S[idx] = v;
    const int nyq = F + L - 1;
    C vc;
    vc.x = v.x; vc.y = -v.y;
    if (n >= L + 1 && n < 2 * L + 1) S[(size_t)m_ext * Np + 2 * L - n] = vc;
    else if (n >= F - 1 && n < nyq) S[(size_t)m_ext * Np + 2 * nyq - n] = vc;
}

template <typename real>
__global__ void __launch_bounds__(1024) k_generic(GenericArgs<real> a) {
    using C = typename cx<real>::type;
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    C *S = a.state + (size_t)b * Tp * Np;
    const real *amp = a.amp + (size_t)b * Tp * Np;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const bool add_self = (a.update == 1);
    // participation masks of the weight rows this launch uses (one 64-bit word per row; see accumulate)
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    unsigned long long *mkb = reinterpret_cast<unsigned long long *>(gsm);
    const bool packed = a.mask_rows > 0;
    const unsigned long long *mk[3] = {nullptr, nullptr, nullptr};
    if (packed) {
        const int RQ = Q * (L + 1);
        const int first = (a.mode == MODE_ONLINE) ? 0 : a.wsel, nsets = (a.mode == MODE_ONLINE) ? 3 : 1;
        for (int i = tid; i < nsets * Qp; i += nthr) {
            const int set = first + i / Qp, row = i % Qp;
            const uint8_t *f = a.w[set].flag + (size_t)row * RQ;
            unsigned long long m = 0;
            for (int x = 0; x < RQ; ++x) m |= (unsigned long long)(f[x] != 0) << x;
            mkb[i] = m;
        }
        for (int set = 0; set < nsets; ++set) mk[first + set] = mkb + (size_t)set * Qp;
        __syncthreads();
    }

    if (a.mode == MODE_BATCH || a.mode == MODE_NOFUTURE || a.mode == MODE_ASYM) {
        const WeightSet<real> ws = a.w[a.wsel];
        const bool asym = (a.mode == MODE_ASYM);
        bool centre = (a.mode == MODE_BATCH);
        int two_sided = (a.mode == MODE_BATCH) ? Q : 1;
        const int nsweeps = a.n_thr;
        int lpi = (F - 1) / sk + 1;  // frames that can sit on one hyperplane
        if (lpi > T) lpi = T;
        const int group = a.group < 1 ? 1 : a.group;
        for (int g0 = 0; g0 < nsweeps; g0 += group) {
            const int ng = (nsweeps - g0 < group) ? nsweeps - g0 : group;
            const int nsteps = F + sk * (T - 1) + D * (ng - 1);
            for (int step = 0; step < nsteps; ++step) {
                for (int idx = tid; idx < ng * lpi; idx += nthr) {
                    const int k = idx / lpi, j = idx - k * lpi;
                    const int u = step - D * k;
                    if (u < 0) continue;
                    int jhi = u / sk;
                    if (jhi > T - 1) jhi = T - 1;
                    const int mm = jhi - j;
                    if (mm < 0) continue;
                    const int c = u - sk * mm;
                    if (c >= F) continue;
                    if (asym) {  // cframe / rframe of lwslib.cpp:1143-1151 for local frame mm
                        two_sided = a.M0 - mm;
                        if (two_sided > Q) two_sided = Q;
                        centre = two_sided >= 1;
                        if (two_sided < 1) two_sided = 1;
                    }
                    update_bin<real>(S, amp, mm + Q - 1, c, centre, two_sided, ws, thr[g0 + k], F, L,
                                     Q, Qp, add_self, a.qdiv, mk[a.wsel]);
                }
                __syncthreads();
            }
        }
    } else if (a.mode == MODE_NOFUTURE_Q4_COMPAT) {
        const WeightSet<real> ws = a.w[a.wsel];
        // bins whose farthest read (m-1)*Np + 2n + L stays inside frame m-1 are independent of frame m
        int n_split = (Np - L + 1) / 2;  // first extended column n with 2n + L >= Np
        if (n_split < L) n_split = L;
        if (n_split > F + L) n_split = F + L;
        for (int s = 0; s < a.n_thr; ++s) {
            const real th = thr[s];
            for (int m = 0; m < T; ++m) {
                for (int n = L + tid; n < n_split; n += nthr)
                    update_bin_nfq4<real>(S, amp, m + Q - 1, n - L, ws, th, F, L, mk[a.wsel]);
                __syncthreads();
                // upper bins: the flat offset of frame m-1 runs into frame m itself, columns 2n - Np +- k <= 2n - Np + L,
                // all below n.  Bins [n0, n1) are independent of each other (and so reproduce the sequential order) as
                // long as every such column lies below n0: n < (n0 + Np - L) / 2.  The remaining range halves each round.
                for (int n0 = n_split; n0 < F + L;) {
                    int n1 = (n0 + Np - L + 1) / 2;
                    if (n1 > F + L) n1 = F + L;
                    // a bin n <= 2L writes its image into column 2L - n; keep such bins out of rounds whose other bins could
                    // read that column through the flat offset (only possible for F <= 3L - 1): run them one by one
                    if (n0 <= 2 * L) n1 = n0 + 1;
                    for (int n = n0 + tid; n < n1; n += nthr)
                        update_bin_nfq4<real>(S, amp, m + Q - 1, n - L, ws, th, F, L, mk[a.wsel]);
                    __syncthreads();
                    n0 = n1;
                }
            }
        }
    } else {  // MODE_ONLINE: TF_RTISI_LA (lwslib.cpp:1432-1491)
        const int LA = a.LA, per = a.n_thr + 1, rps = LA + 1;
        const long nsweeps = (long)T * per;
        // sweep s: frame m = s / per; q = s % per.  q == 0: first estimate of frame m from the past
        // (W_ai, threshold 0, M=1, M0=0).  q >= 1: iteration q-1 over frames max(0,m-LA) .. m, the
        // look-ahead frames with W (M=count, M0=count+1) and frame m with W_af (M=1, M0=1).
        const long t_end = D * (nsweeps - 1) + (long)sk * (T - 1) + F;  // one past the last step
        long s_lo = 0;
        for (long t = 0; t < t_end; ++t) {
            // first sweep that has not finished: end_s = D*s + sk*m(s) + F-1 is increasing in s
            while (s_lo < nsweeps && D * s_lo + (long)sk * (s_lo / per) + F - 1 < t) ++s_lo;
            // sweeps that may have started: D*s + sk*(m(s)-LA) <= t
            long s_hi = s_lo;
            while (s_hi + 1 < nsweeps &&
                   D * (s_hi + 1) + (long)sk * ((s_hi + 1) / per - LA) <= t)
                ++s_hi;
            const int nslots = (int)(s_hi - s_lo + 1) * rps;
            for (int idx = tid; idx < nslots; idx += nthr) {
                const long s = s_lo + idx / rps;
                const int j = idx % rps;
                const int m = (int)(s / per), q = (int)(s % per);
                int first = m - LA;
                if (first < 0) first = 0;
                int rho;
                if (q == 0) { if (j != 0) continue; rho = m; }
                else { rho = first + j; if (rho > m) continue; }
                const long cl = t - D * s - (long)sk * rho;
                if (cl < 0 || cl >= F) continue;
                const int c = (int)cl;
                if (q == 0) {
                    update_bin<real>(S, amp, rho + Q - 1, c, false, 1, a.w[1], (real)0, F, L, Q, Qp,
                                     add_self, a.qdiv, mk[1]);
                } else {
                    int ts = m - rho + 1;
                    if (ts > Q) ts = Q;
                    update_bin<real>(S, amp, rho + Q - 1, c, true, ts, (rho == m) ? a.w[2] : a.w[0],
                                     thr[q - 1], F, L, Q, Qp, add_self, a.qdiv, (rho == m) ? mk[2] : mk[0]);
                }
            }
            __syncthreads();
        }
    }
}

template <typename real>
hipError_t launch_generic(const GenericArgs<real> &a, int B, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    int threads;
    GenericArgs<real> args = a;
    const int sk = a.L + 1;
    if (a.mode == MODE_BATCH || a.mode == MODE_NOFUTURE || a.mode == MODE_ASYM) {
        int lpi = (a.F - 1) / sk + 1;
        if (lpi > a.T) lpi = a.T;
        threads = 1024;
        int group = threads / lpi;
        if (group < 1) group = 1;
        if (group > a.n_thr) group = a.n_thr;
        int need = group * lpi;
        threads = ((need + 63) / 64) * 64;
        if (threads > 1024) threads = 1024;
        if (threads < 64) threads = 64;
        args.group = group;
    } else if (a.mode == MODE_NOFUTURE_Q4_COMPAT) {
        threads = 256;
    } else {
        const int D = a.Q * sk;
        int slots = ((a.F + sk * a.LA) / D + 3) * (a.LA + 1);
        threads = ((slots + 63) / 64) * 64;
        if (threads > 1024) threads = 1024;
        if (threads < 64) threads = 64;
    }
    // flag masks in LDS when a row's Q (L+1) flags fit one word and the rows fit the LDS
    const size_t nsets = (a.mode == MODE_ONLINE) ? 3 : 1;
    size_t lds = 0;
    args.mask_rows = 0;
    if (a.Q * (a.L + 1) <= 64 && nsets * a.Qp * 8 <= 64 * 1024) { args.mask_rows = a.Qp; lds = nsets * a.Qp * 8; }
    hipLaunchKernelGGL(k_generic<real>, dim3(B), dim3(threads), lds, stream, args);
    return hipGetLastError();
}

// =====================================================================================
// batch sweeps on a time-skewed copy of the state: the same wavefront, coalesced
// =====================================================================================
// k_generic keeps the reference's layout, in which the bins of one wavefront step (consecutive frames, each sk bins behind
// the previous one) are Np - sk elements apart: every tap load of a wave touches 64 cache lines, and the step is bound by
// the texture path (57 us per step for Q = 4, 258 us for Q = 8 on 500 x 513).  For batch sweeps -- the mode with the long
// schedules -- the state is first copied to SW[time][frame mod NL], time = sk * frame + column (pad columns included, so the
// Hermitian images stay physical entries exactly as in the reference, lwslib.cpp:362-367): all bins of a step, and each of
// their taps, are consecutive elements of one row.  Same schedule, same arithmetic, same order of operations as
// k_generic's batch mode: bit-identical results.  NL >= Np / sk frames are in flight at a time, so frame m and m + NL
// never share a row.
template <typename real>
__global__ void __launch_bounds__(256) k_to_gskew(const typename cx<real>::type *state, const real *amp,
                                                   typename cx<real>::type *sw, real *aw, int Tp, int Np, int sk, int NL, long rows) {
    using C = typename cx<real>::type;
    const int m = blockIdx.x, b = blockIdx.y;
    const C *src = state + ((size_t)b * Tp + m) * Np;
    const real *asrc = amp + ((size_t)b * Tp + m) * Np;
    C *dst = sw + (size_t)b * rows * NL;
    real *adst = aw + (size_t)b * rows * NL;
    for (int n = threadIdx.x; n < Np; n += blockDim.x) {
        const size_t o = ((size_t)sk * m + n) * NL + (m & (NL - 1));
        dst[o] = src[n];
        adst[o] = asrc[n];
    }
}
template <typename real>
__global__ void __launch_bounds__(256) k_from_gskew(typename cx<real>::type *state, const typename cx<real>::type *sw, int Tp,
                                                     int Np, int sk, int NL, long rows) {
    using C = typename cx<real>::type;
    const int m = blockIdx.x, b = blockIdx.y;
    C *dst = state + ((size_t)b * Tp + m) * Np;
    const C *src = sw + (size_t)b * rows * NL;
    for (int n = threadIdx.x; n < Np; n += blockDim.x) dst[n] = src[((size_t)sk * m + n) * NL + (m & (NL - 1))];
}

template <typename real>
__global__ void __launch_bounds__(1024) k_skew_batch(GenericArgs<real> a, typename cx<real>::type *sw_all, const real *aw_all,
                                                      int NL, long rows) {
    using C = typename cx<real>::type;
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int F = a.F, T = a.T, L = a.L, Q = a.Q, Qp = a.Qp;
    [[maybe_unused]] const int Np = F + 2 * L;
    C *SW = sw_all + (size_t)b * rows * NL;
    const real *AW = aw_all + (size_t)b * rows * NL;
    const real *thr = a.thr + (size_t)b * a.n_thr;
    const int sk = L + 1, D = Q * sk;
    const bool add_self = (a.update == 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    unsigned long long *mkb = reinterpret_cast<unsigned long long *>(gsm);
    const unsigned long long *mk = nullptr;
    const WeightSet<real> ws = a.w[a.wsel];
    if (a.mask_rows > 0) {
        const int RQ = Q * (L + 1);
        for (int row = tid; row < Qp; row += nthr) {
            const uint8_t *f = ws.flag + (size_t)row * RQ;
            unsigned long long m = 0;
            for (int x = 0; x < RQ; ++x) m |= (unsigned long long)(f[x] != 0) << x;
            mkb[row] = m;
        }
        mk = mkb;
        __syncthreads();
    }
    const int nsweeps = a.n_thr;
    int lpi = (F - 1) / sk + 1;  // frames that can sit on one hyperplane
    if (lpi > T) lpi = T;
    const int group = a.group < 1 ? 1 : a.group;
    const int nyq = F + L - 1;
    for (int g0 = 0; g0 < nsweeps; g0 += group) {
        const int ng = (nsweeps - g0 < group) ? nsweeps - g0 : group;
        const int nsteps = F + sk * (T - 1) + D * (ng - 1);
        for (int step = 0; step < nsteps; ++step) {
            for (int idx = tid; idx < ng * lpi; idx += nthr) {
                const int k = idx / lpi, j = idx - k * lpi;
                const int u = step - D * k;
                if (u < 0) continue;
                int jhi = u / sk;
                if (jhi > T - 1) jhi = T - 1;
                // consecutive threads take consecutive frames (ascending): consecutive elements of the rows they touch
                int jlo = (u - (F - 1) + sk - 1) / sk;
                if (jlo < 0) jlo = 0;
                const int mm = jlo + j;
                if (mm > jhi) continue;
                const int c = u - sk * mm;
                if (c >= F) continue;
                const int me = mm + Q - 1, n = c + L;
                const long row = (long)sk * me + n;
                const int lane = me & (NL - 1);
                const real target = AW[row * NL + lane];
                if (!(target > thr[g0 + k])) continue;
                const C acc = accumulate_v<real>(SkewView<C>{SW, row, lane, sk, NL}, c, true, Q, ws, L, Q, Qp, add_self, a.qdiv, mk);
                const real mag = sqrt(acc.x * acc.x + acc.y * acc.y);
                if (!(mag > 0)) continue;
                C v;
                v.x = acc.x * target / mag;
                v.y = acc.y * target / mag;
                SW[row * NL + lane] = v;
                C vc;
                vc.x = v.x; vc.y = -v.y;            // Hermitian images in the pad columns (lwslib.cpp:362-367)
                if (n >= L + 1 && n < 2 * L + 1) SW[(row + 2 * (L - n)) * NL + lane] = vc;
                else if (n >= F - 1 && n < nyq) SW[(row + 2 * (nyq - n)) * NL + lane] = vc;
            }
            __syncthreads();
        }
    }
}

template <typename real>
size_t generic_skew_bytes(int B, int F, int T, int L, int Q, size_t *amp_bytes) {
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), sk = L + 1;
    int NL = 1;
    while (NL * sk < Np) NL *= 2;
    const size_t rows = (size_t)sk * (Tp - 1) + Np;
    if (amp_bytes) *amp_bytes = (size_t)B * rows * NL * sizeof(real);
    return (size_t)B * rows * NL * sizeof(typename cx<real>::type);
}

template <typename real>
hipError_t launch_generic_skewed(const GenericArgs<real> &a, int B, void *sw, void *aw, hipStream_t stream) {
    using C = typename cx<real>::type;
    if (B <= 0) return hipSuccess;
    if (a.mode != MODE_BATCH) return hipErrorInvalidValue;
    const int Np = a.F + 2 * a.L, Tp = a.T + 2 * (a.Q - 1), sk = a.L + 1;
    int NL = 1;
    while (NL * sk < Np) NL *= 2;
    const long rows = (long)sk * (Tp - 1) + Np;
    GenericArgs<real> args = a;
    int lpi = (a.F - 1) / sk + 1;
    if (lpi > a.T) lpi = a.T;
    int threads = 1024;
    int group = threads / lpi;
    if (group < 1) group = 1;
    if (group > a.n_thr) group = a.n_thr;
    threads = ((group * lpi + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    if (threads < 64) threads = 64;
    args.group = group;
    size_t lds = 0;
    args.mask_rows = 0;
    if (a.Q * (a.L + 1) <= 64 && (size_t)a.Qp * 8 <= 64 * 1024) { args.mask_rows = a.Qp; lds = (size_t)a.Qp * 8; }
    hipLaunchKernelGGL(k_to_gskew<real>, dim3(Tp, B), dim3(256), 0, stream, a.state, a.amp, static_cast<C *>(sw), static_cast<real *>(aw),
                       Tp, Np, sk, NL, rows);
    hipLaunchKernelGGL(k_skew_batch<real>, dim3(B), dim3(threads), lds, stream, args, static_cast<C *>(sw),
                       static_cast<const real *>(aw), NL, rows);
    hipLaunchKernelGGL(k_from_gskew<real>, dim3(Tp, B), dim3(256), 0, stream, a.state, static_cast<const C *>(sw), Tp, Np, sk, NL, rows);
    return hipGetLastError();
}
template hipError_t launch_generic_skewed<float>(const GenericArgs<float> &, int, void *, void *, hipStream_t);
template hipError_t launch_generic_skewed<double>(const GenericArgs<double> &, int, void *, void *, hipStream_t);
template size_t generic_skew_bytes<float>(int, int, int, int, int, size_t *);
template size_t generic_skew_bytes<double>(int, int, int, int, int, size_t *);

template hipError_t launch_generic<float>(const GenericArgs<float> &, int, hipStream_t);
template hipError_t launch_generic<double>(const GenericArgs<double> &, int, hipStream_t);

// =====================================================================================
// prep / refresh / extract
// =====================================================================================

template <typename T> struct scalar_of;
template <> struct scalar_of<float2> { using type = float; };
template <> struct scalar_of<double2> { using type = double; };

__device__ __forceinline__ double block_sum(double v, double *red) {
    // deterministic tree reduction over the block (blockDim.x is a power of two <= 256)
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// ExtendSpec + ComputeAmpSpec (lwslib.cpp:15-65 / lws.pyx:146-157,235-240), one block per
// (extended frame, spectrogram).
template <typename real, typename in_cx>
__global__ void __launch_bounds__(256) k_prep(const in_cx *in, typename cx<real>::type *state,
                                               real *amp, double *row_sums, int T, int F, int L,
                                               int Q) {
    using C = typename cx<real>::type;
    __shared__ double red[256];
    const int me = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), nyq = F + L - 1;
    int src = me - (Q - 1);
    const bool real_frame = (src >= 0 && src < T);
    if (src < 0) src = 0;
    if (src > T - 1) src = T - 1;
    const in_cx *row = in + ((size_t)b * T + src) * F;
    C *orow = state + ((size_t)b * Tp + me) * Np;
    real *arow = amp + ((size_t)b * Tp + me) * Np;
    double acc = 0;
    for (int n = threadIdx.x; n < Np; n += blockDim.x) {
        int c = n - L;
        double sgn = 1.0;
        if (c < 0) { c = -c; sgn = -1.0; }                       // below DC: conj of bin -c
        else if (c > F - 1) { c = 2 * (F - 1) - c; sgn = -1.0; }  // above Nyquist
        (void)nyq;
        const in_cx v = row[c];
        const double re = (double)v.x, im = (double)v.y;
        C o;
        o.x = (real)re;
        o.y = (real)(sgn * im);
        orow[n] = o;
        // |S| in fp64.  From float inputs the squares are exact and cannot overflow in fp64: one rounding in the sum, one
        // in the square root (the systolic path's own loader, k_in_to_skew, uses the same form so that both agree bit
        // for bit); complex128 inputs keep hypot's range.
        const double mag = std::is_same<typename scalar_of<in_cx>::type, float>::value ? sqrt(re * re + im * im) : hypot(re, im);
        arow[n] = (real)mag;
        if (real_frame && n >= L && n < F + L) acc += mag;
    }
    const double tot = block_sum(acc, red);
    if (real_frame && threadIdx.x == 0) row_sums[(size_t)b * T + src] = tot;
}

// mean|S| per spectrogram from the per-frame sums, fixed order (deterministic).
__global__ void __launch_bounds__(256) k_mean(const double *row_sums, double *mean_amp, int T, int F) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double acc = 0;
    for (int m = threadIdx.x; m < T; m += blockDim.x) acc += row_sums[(size_t)b * T + m];
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) mean_amp[b] = tot / ((double)T * (double)F);
}

template <typename real>
__global__ void k_scale_thr(const double *thr, const double *mean_amp, real *out, int n) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)b * n + i] = (real)(thr[i] * mean_amp[b]);
}

// What a fresh extspec() + abs() + mean() on the current state would produce (lws.pyx:235-240):
// edge-pad frames become copies of the current first / last frame, AmpSpec = |state|.
template <typename real>
__global__ void __launch_bounds__(256) k_refresh(typename cx<real>::type *state, real *amp,
                                                  double *row_sums, int T, int F, int L, int Q) {
    using C = typename cx<real>::type;
    __shared__ double red[256];
    const int me = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    int src = me - (Q - 1);
    const bool real_frame = (src >= 0 && src < T);
    if (src < 0) src = 0;
    if (src > T - 1) src = T - 1;
    const C *row = state + ((size_t)b * Tp + src + Q - 1) * Np;
    C *orow = state + ((size_t)b * Tp + me) * Np;
    real *arow = amp + ((size_t)b * Tp + me) * Np;
    double acc = 0;
    for (int n = threadIdx.x; n < Np; n += blockDim.x) {
        const C v = row[n];
        if (!real_frame) orow[n] = v;
        const double mag = hypot((double)v.x, (double)v.y);
        arow[n] = (real)mag;
        if (real_frame && n >= L && n < F + L) acc += mag;
    }
    const double tot = block_sum(acc, red);
    if (real_frame && threadIdx.x == 0) row_sums[(size_t)b * T + src] = tot;
}

// CopySpec (lwslib.cpp:47-57 / lws.pyx:256).
template <typename real, typename out_cx>
__global__ void __launch_bounds__(256) k_extract(const typename cx<real>::type *state, out_cx *out,
                                                  const out_cx *orig, int T, int F, int L, int Q) {
    using C = typename cx<real>::type;
    using oreal = typename scalar_of<out_cx>::type;
    const int m = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    const C *row = state + ((size_t)b * Tp + m + Q - 1) * Np + L;
    out_cx *orow = out + ((size_t)b * T + m) * F;
    const out_cx *grow = orig ? orig + ((size_t)b * T + m) * F : nullptr;
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        const C v = row[c];
        out_cx o;
        o.x = (oreal)v.x;
        o.y = (oreal)v.y;
        if (grow) {
            const out_cx g = grow[c];
            // a bin the sweeps never changed still holds the rounded original: hand back the original
            if ((real)g.x == v.x && (real)g.y == v.y) o = g;
        }
        orow[c] = o;
    }
}

// Consistency-residual proxy: since create_weights subtracts 1 from the
// zero-lag weight (lws.pyx:177), acc + W[.,0,0]*S is the truncated (F(S)-S)[m,n].  One block per
// (frame, spectrogram) writes sum|acc + w00 S|^2 and sum|S|^2 of that frame; reduced in fixed order.
template <typename real>
__global__ void __launch_bounds__(256) k_residual_rows(const typename cx<real>::type *state,
                                                        WeightSet<real> ws, double *rows, int T,
                                                        int F, int L, int Q, int Qp) {
    using C = typename cx<real>::type;
    __shared__ double red[256];
    const int m = blockIdx.x, b = blockIdx.y;
    const int Np = F + 2 * L, Tp = T + 2 * (Q - 1), RQ = Q * (L + 1);
    const C *row = state + ((size_t)b * Tp + m + Q - 1) * Np + L;
    double e = 0, p = 0;
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        const C a = accumulate<real>(row + c, c, true, Q, ws, Np, L, Q, Qp, false, (real)1);
        const C w0 = ws.w[(size_t)(c % Qp) * RQ];
        const C s = row[c];
        const double rx = (double)a.x + (double)w0.x * s.x - (double)w0.y * s.y;
        const double ry = (double)a.y + (double)w0.x * s.y + (double)w0.y * s.x;
        e += rx * rx + ry * ry;
        p += (double)s.x * s.x + (double)s.y * s.y;
    }
    const double et = block_sum(e, red);
    const double pt = block_sum(p, red);
    if (threadIdx.x == 0) {
        rows[((size_t)b * T + m) * 2] = et;
        rows[((size_t)b * T + m) * 2 + 1] = pt;
    }
}
__global__ void __launch_bounds__(256) k_residual_sum(const double *rows, double *out, int T) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double e = 0, p = 0;
    for (int m = threadIdx.x; m < T; m += blockDim.x) {
        e += rows[((size_t)b * T + m) * 2];
        p += rows[((size_t)b * T + m) * 2 + 1];
    }
    const double et = block_sum(e, red);
    const double pt = block_sum(p, red);
    if (threadIdx.x == 0) { out[2 * b] = et; out[2 * b + 1] = pt; }
}

template <typename real>
hipError_t launch_residual(const typename cx<real>::type *state, WeightSet<real> ws, double *rows,
                           double *out, int B, int T, int F, int L, int Q, int Qp,
                           hipStream_t stream) {
    hipLaunchKernelGGL(k_residual_rows<real>, dim3(T, B), dim3(256), 0, stream, state, ws, rows, T, F,
                       L, Q, Qp);
    hipLaunchKernelGGL(k_residual_sum, dim3(B), dim3(256), 0, stream, rows, out, T);
    return hipGetLastError();
}
template hipError_t launch_residual<float>(const float2 *, WeightSet<float>, double *, double *, int, int, int, int, int, int, hipStream_t);
template hipError_t launch_residual<double>(const double2 *, WeightSet<double>, double *, double *, int, int, int, int, int, int, hipStream_t);

template <typename real, typename in_cx>
hipError_t launch_prep(const in_cx *in, typename cx<real>::type *state, real *amp, double *row_sums,
                       double *mean_amp, int B, int T, int F, int L, int Q, hipStream_t stream) {
    const int Tp = T + 2 * (Q - 1);
    hipLaunchKernelGGL((k_prep<real, in_cx>), dim3(Tp, B), dim3(256), 0, stream, in, state, amp,
                       row_sums, T, F, L, Q);
    hipLaunchKernelGGL(k_mean, dim3(B), dim3(256), 0, stream, row_sums, mean_amp, T, F);
    return hipGetLastError();
}

template <typename real>
hipError_t launch_scale_thresholds(const double *thr, const double *mean_amp, real *out, int B, int n,
                                   hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scale_thr<real>, dim3((n + 127) / 128, B), dim3(128), 0, stream, thr,
                       mean_amp, out, n);
    return hipGetLastError();
}

template <typename real>
hipError_t launch_refresh(typename cx<real>::type *state, real *amp, double *row_sums,
                          double *mean_amp, int B, int T, int F, int L, int Q, hipStream_t stream) {
    const int Tp = T + 2 * (Q - 1);
    // real frames first (they are the sources), then the pad frames
    hipLaunchKernelGGL(k_refresh<real>, dim3(Tp, B), dim3(256), 0, stream, state, amp, row_sums, T, F,
                       L, Q);
    hipLaunchKernelGGL(k_mean, dim3(B), dim3(256), 0, stream, row_sums, mean_amp, T, F);
    return hipGetLastError();
}

template <typename real, typename out_cx>
hipError_t launch_extract(const typename cx<real>::type *state, out_cx *out, const out_cx *orig,
                          int B, int T, int F, int L, int Q, hipStream_t stream) {
    hipLaunchKernelGGL((k_extract<real, out_cx>), dim3(T, B), dim3(256), 0, stream, state, out, orig,
                       T, F, L, Q);
    return hipGetLastError();
}

#define LWS_INST_PREP(real, in_cx)                                                                  \
    template hipError_t launch_prep<real, in_cx>(const in_cx *, cx<real>::type *, real *, double *, \
                                                 double *, int, int, int, int, int, hipStream_t);
LWS_INST_PREP(float, double2)
LWS_INST_PREP(float, float2)
LWS_INST_PREP(double, double2)
template hipError_t launch_scale_thresholds<float>(const double *, const double *, float *, int, int, hipStream_t);
template hipError_t launch_scale_thresholds<double>(const double *, const double *, double *, int, int, hipStream_t);
template hipError_t launch_refresh<float>(float2 *, float *, double *, double *, int, int, int, int, int, hipStream_t);
template hipError_t launch_refresh<double>(double2 *, double *, double *, double *, int, int, int, int, int, hipStream_t);
#define LWS_INST_EXTRACT(real, out_cx)                                                         \
    template hipError_t launch_extract<real, out_cx>(const cx<real>::type *, out_cx *,         \
                                                     const out_cx *, int, int, int, int, int,  \
                                                     hipStream_t);
LWS_INST_EXTRACT(float, double2)
LWS_INST_EXTRACT(float, float2)
LWS_INST_EXTRACT(double, double2)

}  // namespace lws
