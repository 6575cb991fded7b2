// lws_systolic.hip -- placeholder until the systolic kernel lands: reports "not applicable".
#include "lws_systolic.h"
namespace lws {
hipError_t systolic_build(SystolicPlan &, int, int, int, int, const double *const[3]) { return hipSuccess; }
void systolic_release(SystolicPlan &) {}
bool systolic_supports(const SystolicPlan &, int, int) { return false; }
const char *systolic_name(const SystolicPlan &sp) { return sp.name; }
hipError_t launch_systolic(SystolicPlan &, int, float2 *, const float *, const float *, int, int, int,
                           hipStream_t, int *, hipEvent_t, hipEvent_t) { return hipErrorNotSupported; }
}  // namespace lws
