// lws_online.h -- LDS-resident fp32 engine for the online driver (lws_online.hip).  Internal, not part of the ABI.
#pragma once
#include "lws_common.h"

namespace lws {

// true if launch_online_lds can run this shape: all three tensors with the common twiddle structure (tw_P, tw_s) that
// weights_twiddle finds -- static eighth turns (P = Q in {2,4,8}, s = 1: every layout) or a table (Q in 3..8, any P <= 512: the
// fourth layout) -- L <= 5 (L = 5 for the first three layouts), the window of frames the sweeps in flight need fits the LDS ring;
// otherwise the caller uses the generic engine.
bool online_lds_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update, int tw_P, int tw_s, bool table);
// [P + 3][TQ] complex twiddles for the table variant, TQ = 4 (Q <= 4) or 8 (out: 2 (P + 3) TQ floats)
void online_twiddle_table(int P, int s, int Q, float *out);
// do the twiddles exp(2 pi j p r s / P) need no table (eighth turns of Q in {2,4,8})?
bool online_static_twiddles(int Q, int tw_P, int tw_s);

// Same contract as launch_generic<float> with mode == MODE_ONLINE.  tw_table_dev: the uploaded online_twiddle_table (table variant)
hipError_t launch_online_lds(const GenericArgs<float> &a, int B, int tw_P, int tw_s, const float *tw_table_dev, hipStream_t stream);

}  // namespace lws
