/* s_out = batch_lws(s_in, weights, thresholds)      -- GPU gateway, syntax of the reference's matlab/batch_lws.cpp:2
 *   s_in        Nreal x T (x B) double, real or complex: non-negative frequencies of the STFT
 *   weights     (L+1) x Q x Q complex double from create_weights.m
 *   thresholds  real vector; its length is the number of sweeps, each entry is scaled by mean(abs(s_in)) per spectrogram
 * Build:  mex batch_lws.cpp -I<repo>/include -L<repo>/lws_amd -llws_hip
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
    if (lws_batch_lws(plan, LWS_W, S.z.data(), S.z.data(), S.B, S.T, mxGetPr(prhs[2]),
                      (int)mxGetNumberOfElements(prhs[2])) != LWS_OK) {
        mexPrintf("lws: %s\n", lws_last_error());
        return;
    }
    plhs[0] = lwsmex::write_spec(S);
}
