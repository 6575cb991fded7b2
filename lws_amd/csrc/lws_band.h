// lws_band.h -- interface of the band engine (lws_band.hip): batch sweeps of the shapes no systolic build takes.
#pragma once
#include "lws_band_core.h"
#include "lws_common.h"

#include <vector>

namespace lws {

// How a call runs on the band engine (band_plan): the geometry (lws_band_core.h), the twiddle (Pt, s) of the weight tensor, the
// sweep slots per workgroup, the instantiation (LT >= L, QT >= Q), the scratch it needs.
struct BandPlan {
    band::Geom g;
    int NS, LT, QT, Pt, s, L, chunk, helpers;   // helpers: helper waves per sweep slot (exact builds: lws_band_core.h, Split)
    bool fp64;
    size_t state_bytes, amp_bytes;   // scratch: the time-skewed state of `chunk` spectrograms; the magnitudes
};
// What the band engine can run: MODE_BATCH, update == 2, 2 <= Q <= 16, L <= 10, F >= 2 LT + 7 (LT = 5 for L <= 5, else 10), a weight
// tensor with create_weights' twiddle structure (lws.pyx:160-181; weights_twiddle(), any twiddle period whose table leaves room for a ring in the LDS -- for an fp64 plan
// a summarised tensor (Qp == Q) whose rows are twiddle images of row 0 to 1e-13, so that the results are the reference's to rounding) and a frame short enough for one sweep
// slot's ring in the LDS (Q F complex values: e.g. Q = 8 at 1025 bins, Q = 16 at 513 bins in fp32; half that in fp64).
// W: the plan's tensor on the host (complex128 interleaved, [Qp][Q][L+1]).  False: the caller uses the generic engine.
bool band_plan(bool fp64, int B, int F, int T, int L, int Q, int Qp, int update, int n_thr, const double *W, BandPlan *out);
const char *band_name(const BandPlan &bp);   // "band_fp32" / "band_fp64"
// The two tables a kernel reads (lws_band_host.h: tables -- [Q][LT+1] weights, then [Pt][Q-1] twiddles), in the plan's arithmetic
// type: what the caller keeps on the device for the plan's lifetime (they depend on W, LT and the precision only) and hands to
// launch_band.
std::vector<unsigned char> band_tables(const BandPlan &bp, const double *W_host);
// Runs a.n_thr batch sweeps on the extended buffers a.state / a.amp (reference layout), in place.  Same sweeps in the reference's
// order; a bin's sum is taken in another order than lwslib.cpp:297-354 takes it (scatter form), so results agree with
// launch_generic<real> to rounding, not bit for bit.  ev0 / ev1 (may be null) bracket the update kernels.
template <typename real>
hipError_t launch_band(const BandPlan &bp, const GenericArgs<real> &a, const void *tables_dev, int B, void *skew_state, void *skew_amp,
                       hipStream_t stream, int *launches, hipEvent_t ev0, hipEvent_t ev1);

}  // namespace lws
