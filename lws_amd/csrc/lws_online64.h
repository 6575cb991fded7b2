// lws_online64.h -- interface of the fp64 online LDS engine (lws_online64.hip).
#pragma once
#include "lws_common.h"

namespace lws {

// true if launch_online64 can run this shape: summarised weights (Qp == Q, Q in {2, 3, 4, 8}), L = 5, update == 2, and the frames the
// sweeps in flight need fit the LDS as fp64 rows (frames of up to ~600 bins); otherwise the caller uses the generic engine.
bool online64_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update);
// Same contract and the same BITS as launch_generic<double> with mode == MODE_ONLINE.
hipError_t launch_online64(const GenericArgs<double> &a, int B, hipStream_t stream);
// "online_lds_fp64"; "online_lds_fp64_1w" when LWS_ONLINE64_ONE_WAVE selects the one-wave kernel (comparison runs; read on every launch)
const char *online64_name();

}  // namespace lws
