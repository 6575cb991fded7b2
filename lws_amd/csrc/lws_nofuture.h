// lws_nofuture.h -- LDS-resident engine for the no-future sweeps (lws_nofuture.hip).  Internal, not part of the ABI.
#pragma once
#include "lws_common.h"

namespace lws {

// rows: the weight rows the kernel keeps in LDS -- Q for a summarised tensor (Qp == Q), weights_row_period() for a general one
// (Qp == N rows that repeat; 0 if they do not).
// true if launch_nofuture_lds can run this shape (periodic weight rows, the ring of Q + 1 frames fits the LDS);
// otherwise the caller uses the generic engine.
bool nofuture_lds_supports(int F, int T, int L, int Q, int Qp, int rows);
int weights_row_period(const double *W, int Qp, int Q, int L, int pmax);

// Same contract as launch_generic<float> with mode == MODE_NOFUTURE or MODE_NOFUTURE_Q4_COMPAT; summarised tensors: bit-identical
// results (general ones: the same weights up to 1e-9 relative before they are rounded to fp32).
hipError_t launch_nofuture_lds(const GenericArgs<float> &a, int B, int rows, hipStream_t stream);

// fp64 plans (round 5): the one-lane-per-bin variant in double, summarised tensors -- launch_generic<double>'s results bit for bit.
bool nofuture_lds64_supports(int F, int T, int L, int Q, int Qp, int rows);
hipError_t launch_nofuture_lds64(const GenericArgs<double> &a, int B, int rows, hipStream_t stream);

}  // namespace lws
