// lws_sys64.h -- interface of the fp64 systolic batch engine (lws_sys64.hip).
#pragma once
#include "lws_common.h"

namespace lws {

// Batch sweeps (MODE_BATCH, update == 2) of an fp64 plan with Q in {2, 4}, L = 5 and
// frames short enough for at least one sweep slot's ring in the LDS (up to ~620 bins with 64 frames in flight, ~1070 with 128, ~2090 with 256).
// W: the plan's weight tensor on the host (complex128 interleaved, [Qp][Q][L+1]); its rows must be the quarter-turn images of
// row 0 that create_weights (lws.pyx:160-181) produces, to 1e-13.
bool sys64_supports(int F, int T, int L, int Q, int Qp, int update, const double *W);
// Scratch of a call: the time-skewed state (return value), the magnitudes in the same addressing.
size_t sys64_bytes(int B, int F, int T, int Q, size_t *amp_bytes);
const char *sys64_name(int F, int T, int Q);   // "..._wide" / "..._xwide": 128 / 256 frames in flight, two / four waves per sweep slot
// Diagnostics (tests): out = {rows allocated per workgroup, highest row the prefetch reads, highest row written, gap}
bool sys64_layout(int F, int T, int Q, long out[4]);
// Runs a.n_thr batch sweeps on the extended buffers a.state / a.amp (reference layout), in place.  Same results as
// launch_generic<double> up to the rounding of a different summation order.  ev0 / ev1 (may be null) bracket the update kernels.
hipError_t launch_sys64(const GenericArgs<double> &a, const double *W_host, int B, void *skew_state, void *skew_amp, hipStream_t stream,
                        int *launches, hipEvent_t ev0, hipEvent_t ev1);

}  // namespace lws
