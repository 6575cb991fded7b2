/* s_out = online_lws(s_in, weights, weights_asym_init, weights_asym_full, thresholds, LA)
 *                                                   -- GPU gateway, syntax of the reference's matlab/online_lws.cpp:2,25-30
 * Frame-by-frame reconstruction with LA look-ahead frames (TF_RTISI_LA, lwslib.cpp:1424-1492): each new frame is
 * initialised from the past with weights_asym_init, then length(thresholds) rounds re-sweep the LA previous frames
 * with `weights` and the newest with weights_asym_full.
 * Build:  mex online_lws.cpp -I<repo>/include -L<repo>/lws_amd -llws_hip
 */
#include "lws_mex_common.h"

static lwsmex::PlanCache g_cache;
static void release() { g_cache.drop(); }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs < 6) {
        mexPrintf("lws: not enought inputs\n");
        return;
    }
    lwsmex::Spec S;
    lwsmex::Weights W, Wai, Waf;
    if (!lwsmex::read_spec(prhs[0], S) || !lwsmex::read_weights(prhs[1], W, "weights") ||
        !lwsmex::read_weights(prhs[2], Wai, "weights_asym_init") ||
        !lwsmex::read_weights(prhs[3], Waf, "weights_asym_full"))
        return;
    if (Wai.w.size() != W.w.size() || Waf.w.size() != W.w.size()) {
        mexPrintf("lws: the three weight arrays must have the same size.\n");
        return;
    }
    if (!lwsmex::real_vector(prhs[4])) {
        mexPrintf("lws: please provide a 1-D list of phase update thresholds.\n");
        return;
    }
    if (!lwsmex::full_double(prhs[5]) || mxGetNumberOfElements(prhs[5]) != 1) {
        mexPrintf("Number of look-ahead frames is not a real scalar.\n");
        return;
    }
    const int LA = (int)mxGetScalar(prhs[5]);
    if (nlhs < 1) return;
    mexAtExit(release);
    lws_plan *plan = g_cache.get(S.F, W, &Wai, &Waf);
    if (!plan) return;
    /* Qfloat = Q as online_lws.cpp:160 passes it */
    if (lws_online_lws(plan, S.z.data(), S.z.data(), S.B, S.T, mxGetPr(prhs[4]), (int)mxGetNumberOfElements(prhs[4]),
                       LA, (double)W.Q) != LWS_OK) {
        mexPrintf("lws: %s\n", lws_last_error());
        return;
    }
    plhs[0] = lwsmex::write_spec(S);
}
