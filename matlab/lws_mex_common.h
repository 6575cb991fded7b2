/* Shared plumbing of the three MATLAB gateways in this directory (batch_lws, nofuture_lws, online_lws), which keep
 * the call syntax of the reference's matlab/batch_lws.cpp:2, nofuture_lws.cpp:2 and online_lws.cpp:2 and run the
 * updates on the GPU through the C ABI of include/lws_hip.h instead of lwslib.cpp on the host.
 *
 * What the reference gateways do per call (batch_lws.cpp:78-150): build the flag mask, extend the spectrogram, take
 * |S| and its mean, scale the thresholds, sweep, copy the centre back.  All of that lives behind lws_plan_create /
 * lws_*_lws here; the gateway only converts between MATLAB's split-complex column-major arrays and the interleaved
 * [B][T][F] complex128 layout of the ABI.  A MATLAB Nreal x T matrix (column-major) *is* a [T][Nreal] row-major
 * array and an (L+1) x Q x Q' weight array *is* W[Q'][Q][L+1], so no transposition is needed.
 *
 * Extension over the reference: s_in may be Nreal x T x B; the B spectrograms are processed in one device batch.
 * Environment: LWS_MEX_FP64=1 selects the fp64 plan (reference arithmetic), LWS_MEX_DEVICE=<n> the GPU.
 * Plans are cached per (shape, weights) across calls and released by mexAtExit.
 */
#ifndef LWS_MEX_COMMON_H_
#define LWS_MEX_COMMON_H_

#include <cstdlib>
#include <cstring>
#include <vector>

#include "mex.h"
#include "matrix.h"
#include "lws_hip.h"

namespace lwsmex {

inline bool full_double(const mxArray *a) { return mxIsDouble(a) && !mxIsSparse(a); }

inline bool real_vector(const mxArray *a) {
    return full_double(a) && !mxIsComplex(a) && mxGetNumberOfDimensions(a) == 2 && (mxGetM(a) == 1 || mxGetN(a) == 1);
}

struct Weights {
    int K1 = 0, Q = 0, Qp = 0;
    std::vector<double> w;  // interleaved complex, (L+1) fastest
};

/* (L+1) x Q x Q' complex (or real) double array -> interleaved copy.  A 2-D array is Q' == 1 only if Q == 1, which no
 * window produces, so three dimensions are required exactly as the reference requires them (batch_lws.cpp:43-47). */
inline bool read_weights(const mxArray *a, Weights &out, const char *what) {
    if (!full_double(a) || mxGetNumberOfDimensions(a) != 3) {
        mexPrintf("lws: %s should be a full 3-dimensional double array.\n", what);
        return false;
    }
    const mwSize *d = mxGetDimensions(a);
    out.K1 = (int)d[0];
    out.Q = (int)d[1];
    out.Qp = (int)d[2];
    const size_t n = (size_t)out.K1 * out.Q * out.Qp;
    const double *re = mxGetPr(a), *im = mxIsComplex(a) ? mxGetPi(a) : nullptr;
    out.w.resize(2 * n);
    for (size_t i = 0; i < n; ++i) {
        out.w[2 * i] = re[i];
        out.w[2 * i + 1] = im ? im[i] : 0.0;
    }
    return true;
}

/* one cached plan per gateway */
struct PlanCache {
    lws_plan *plan = nullptr;
    int F = 0;
    unsigned flags = 0;
    std::vector<double> key;  // the concatenated weight tensors the plan was built from
    int K1 = 0, Q = 0, Qp = 0;

    void drop() {
        if (plan) lws_plan_destroy(plan);
        plan = nullptr;
    }

    lws_plan *get(int F_, const Weights &W, const Weights *Wai, const Weights *Waf) {
        unsigned fl = LWS_NOFUTURE_Q4_COMPAT;  // the gateways call NoFuture_LWSQ4 for Q == 4 (nofuture_lws.cpp:133-135)
        const char *p64 = std::getenv("LWS_MEX_FP64");
        if (p64 && p64[0] == '1') fl |= LWS_PRECISION_FP64;
        std::vector<double> k(W.w);
        if (Wai) k.insert(k.end(), Wai->w.begin(), Wai->w.end());
        if (Waf) k.insert(k.end(), Waf->w.begin(), Waf->w.end());
        if (plan && F == F_ && flags == fl && K1 == W.K1 && Q == W.Q && Qp == W.Qp && key == k) return plan;
        drop();
        const char *dev = std::getenv("LWS_MEX_DEVICE");
        if (lws_plan_create(&plan, dev ? std::atoi(dev) : 0, F_, W.K1 - 1, W.Q, W.Qp, W.w.data(),
                            Wai ? Wai->w.data() : nullptr, Waf ? Waf->w.data() : nullptr, fl) != LWS_OK) {
            mexPrintf("lws: %s\n", lws_last_error());
            plan = nullptr;
            return nullptr;
        }
        F = F_; flags = fl; K1 = W.K1; Q = W.Q; Qp = W.Qp;
        key.swap(k);
        return plan;
    }
};

struct Spec {
    int F = 0, T = 0, B = 1;
    std::vector<double> z;  // [B][T][F] interleaved
};

/* s_in: Nreal x T (x B), real or complex.  Same checks and messages as batch_lws.cpp:33-36,73-76. */
inline bool read_spec(const mxArray *a, Spec &s) {
    const mwSize nd = mxGetNumberOfDimensions(a);
    if (!full_double(a) || nd > 3) {
        mexPrintf("lws: spectrogram must be full 2-D double matrix (or Nreal x T x B stack).\n");
        return false;
    }
    const mwSize *d = mxGetDimensions(a);
    s.F = (int)d[0];
    s.T = (int)d[1];
    s.B = nd == 3 ? (int)d[2] : 1;
    if (s.F % 2 == 0) {
        mexPrintf("Please only include non-negative frequencies in the input spectrogram.\n");
        return false;
    }
    const size_t n = (size_t)s.F * s.T * s.B;
    const double *re = mxGetPr(a), *im = mxIsComplex(a) ? mxGetPi(a) : nullptr;
    s.z.resize(2 * n);
    for (size_t i = 0; i < n; ++i) {
        s.z[2 * i] = re[i];
        s.z[2 * i + 1] = im ? im[i] : 0.0;
    }
    return true;
}

inline mxArray *write_spec(const Spec &s) {
    mwSize dims[3] = {(mwSize)s.F, (mwSize)s.T, (mwSize)s.B};
    mxArray *out = mxCreateNumericArray(s.B > 1 ? 3 : 2, dims, mxDOUBLE_CLASS, mxCOMPLEX);
    double *re = mxGetPr(out), *im = mxGetPi(out);
    const size_t n = (size_t)s.F * s.T * s.B;
    for (size_t i = 0; i < n; ++i) {
        re[i] = s.z[2 * i];
        im[i] = s.z[2 * i + 1];
    }
    return out;
}

}  // namespace lwsmex
#endif
