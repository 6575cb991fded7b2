// lws_online.h -- LDS-resident fp32 engine for the online driver (lws_online.hip).  Internal, not part of the ABI.
#pragma once
#include "lws_common.h"

namespace lws {

// true if launch_online_lds can run this shape (summarised weights with the twiddle structure of create_weights in all
// three tensors, L <= 5 (L = 5 for the first three layouts), Q in {2,4,8}, the window of frames the sweeps in flight need fits the LDS ring); otherwise the
// caller uses the generic engine.
bool online_lds_supports(int F, int T, int L, int Q, int Qp, int LA, int n_thr, int update, bool twiddle_structure);
// host check on one complex128 weight tensor [Qp][Q][L+1]: W[p][r][k] == W[0][r][k] exp(2 pi j p r / Q)
bool weights_have_twiddle_structure(const double *W, int Q, int Qp, int L);

// Same contract as launch_generic<float> with mode == MODE_ONLINE.
hipError_t launch_online_lds(const GenericArgs<float> &a, int B, hipStream_t stream);

}  // namespace lws
