/* s_out = nofuture_lws(s_in, weights, thresholds)   -- GPU gateway, syntax of the reference's matlab/nofuture_lws.cpp:2
 * Sweeps that use past frames only (run_lws.m passes the asymmetric "init" weights).  For Q == 4 the plan reproduces
 * the addressing of NoFuture_LWSQ4 (lwslib.cpp:559-594), which is what the reference gateway dispatches to.
 * Build:  mex nofuture_lws.cpp -I<repo>/include -L<repo>/lws_amd -llws_hip
 */
#include "lws_mex_common.h"

static lwsmex::PlanCache g_cache;
static void release() { g_cache.drop(); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 3) {
        mexPrintf("lws: not enought inputs\n");
        return;
    }
    lwsmex::Spec S;
    lwsmex::Weights W;
    if (!lwsmex::read_spec(prhs[0], S) || !lwsmex::read_weights(prhs[1], W, "weights")) return;
    if (!lwsmex::real_vector(prhs[2])) {
        mexPrintf("lws: please provide a 1-D list of phase update thresholds.\n");
        return;
    }
    if (nlhs < 1) return;
    mexAtExit(release);
    lws_plan *plan = g_cache.get(S.F, W, nullptr, nullptr);
    if (!plan) return;
    if (lws_nofuture_lws(plan, LWS_W, S.z.data(), S.z.data(), S.B, S.T, mxGetPr(prhs[2]),
                         (int)mxGetNumberOfElements(prhs[2])) != LWS_OK) {
        mexPrintf("lws: %s\n", lws_last_error());
        return;
    }
    plhs[0] = lwsmex::write_spec(S);
}
