// lws_nofuture.h -- LDS-resident fp32 engine for the no-future sweeps (lws_nofuture.hip).  Internal, not part of the ABI.
#pragma once
#include "lws_common.h"

namespace lws {

// true if launch_nofuture_lds can run this shape (summarised weights, the ring of Q + 1 frames fits the LDS);
// otherwise the caller uses the generic engine.
bool nofuture_lds_supports(int F, int T, int L, int Q, int Qp);

// Same contract as launch_generic<float> with mode == MODE_NOFUTURE or MODE_NOFUTURE_Q4_COMPAT; bit-identical results.
hipError_t launch_nofuture_lds(const GenericArgs<float> &a, int B, hipStream_t stream);

}  // namespace lws
