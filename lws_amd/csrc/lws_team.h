// lws_team.h -- interface of the team engine (lws_team.hip): the generic engine's wavefront schedule with the taps of a bin
// spread over a team of lanes.
#pragma once
#include "lws_common.h"

namespace lws {

// true if launch_team can run this stage: mode MODE_ONLINE or MODE_NOFUTURE, any weights (summarised or general), any Q, L;
// the units a wavefront step holds must fit one workgroup with at least two lanes each (else the generic engine is as good).
bool team_supports(int mode, int F, int T, int L, int Q, int Qp, int LA, int n_thr);
// Same contract as launch_generic (same sweeps, same order of bins, same arithmetic per tap); a bin's sum is taken as per-lane
// partial sums in a fixed order, so results agree with the generic engine to rounding, not bit for bit.
template <typename real>
hipError_t launch_team(const GenericArgs<real> &a, int B, hipStream_t stream);
// true if the online stage of this shape runs with its moving window in LDS (k_team_online_ring) and with at least 8 lanes per bin: the
// case in which the team engine is faster than the fp64 LDS engine's Q = 8 kernel (lws_capi.hip: run_stage)
bool team_online_in_lds(bool fp64, int F, int T, int L, int Q, int Qp, int LA, int n_thr);
// true if launch_team runs the online stage of such a plan on the order-exact kernel (k_team_online_ordered: the generic engine's bits):
// fp64 plans unless LWS_TEAM_FP64=1, fp32 plans with LWS_TEAM_ORDERED=1; ..._fits: its increments fit the LDS
bool team_online_is_ordered(bool fp64);
bool team_ordered_fits(int F, int T, int L, int Q, int LA, int n_thr, bool fp64);
// lanes per bin the launcher chooses for this stage (reported by the tests / tools)
int team_lanes(int mode, int F, int T, int L, int Q, int LA, int n_thr);

}  // namespace lws
