/*
 * lwslib_compat.h -- the native interface of the reference's core (lwslib/lwslib.h:6-26), served by the MI355X
 * engine.  Same function names, argument order, C++ linkage and in-place semantics, so that code written against
 * the reference's header -- its mex gateways matlab/{batch,online,nofuture}_lws.cpp, the `cdef extern` block of
 * python/lwslib.pxd -- compiles and links unchanged against liblws_hip.so:  #include "lwslib_compat.h"  instead of
 * "lwslib.h" (or copy it under that name on the include path).
 *
 * Every kernel call is one fp64 sweep executed on the GPU by the order-exact generic engine (upload the frames the
 * call can touch, sweep, download the updated frames); the three helpers are plain host loops as in the reference.
 * These shims exist for source compatibility.  Throughput comes from the batched entry points of lws_hip.h, which
 * keep spectrograms on the device across sweeps.
 *
 * Conventions (as in the reference): all buffers caller-owned; Sr/Si/AmpSpec address extended row 0 of the call
 * (row pitch Nreal + 2L); frames Q-1 .. M+Q-2 are updated; wr/wi/w_flag are [Qp][Q][L+1] with Qp = Q for the
 * Q2/Q4/anyQ kernels and Qp = 2(Nreal-1) for the fractionalQ ones.  Differences from the reference, both deliberate:
 * the fractionalQ kernels index the weight row of the DC bin periodically (row N = row 0) where the reference reads
 * one row past the end (lwslib.cpp:408,711,1308); failures of the HIP runtime are reported through
 * lwslib_compat_last_error() (the reference's kernels cannot fail) and leave the buffers untouched.
 */
#ifndef LWSLIB_COMPAT_H_INCLUDED
#define LWSLIB_COMPAT_H_INCLUDED

#include <math.h>

/* helpers -- lwslib.h:6-8, lwslib.cpp:15-65 */
void ExtendSpec(double *ExtSr, double *ExtSi, double *InSr, double *InSi, int Nreal, int M, int L, int Q);
void CopySpec(double *ExtSr, double *ExtSi, double *InSr, double *InSi, int Nreal, int M, int L, int Q);
void ComputeAmpSpec(double *Sr, double *Si, double *AmpSpec, int size);

/* batch sweeps -- lwslib.h:10-13, lwslib.cpp:72-467 */
void LWSQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
           int Nreal, int M, int L, double threshold);
void LWSQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
           int Nreal, int M, int L, double threshold);
void LWSanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
             int Nreal, int M, int L, int Q, double threshold);
void LWSfractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                    int Nreal, int M, int L, int Q, double threshold);

/* sweeps that use past frames only -- lwslib.h:15-18, lwslib.cpp:473-764.
 * NoFuture_LWSQ4 reproduces the reference's addressing (flat offset (m-r)*Np + 2n +- k, lwslib.cpp:559-594). */
void NoFuture_LWSQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                    int Nreal, int M, int L, double threshold);
void NoFuture_LWSQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                    int Nreal, int M, int L, double threshold);
void NoFuture_LWSanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                      int Nreal, int M, int L, int Q, double threshold);
void NoFuture_LWSfractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                             int Nreal, int M, int L, int Q, double threshold);

/* sweeps over M frames that may read M0 frames to their right -- lwslib.h:20-23, lwslib.cpp:776-1421 */
void Asym_UpdatePhaseQ2(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                        int Nreal, int M, int M0, int L, double threshold, int update);
void Asym_UpdatePhaseQ4(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                        int Nreal, int M, int M0, int L, double threshold, int update);
void Asym_UpdatePhaseanyQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                          int Nreal, int M, int M0, int L, int Q, double threshold, int update);
void Asym_UpdatePhasefractionalQ(double *Sr, double *Si, double *wr, double *wi, int *w_flag, double *AmpSpec,
                                 int Nreal, int M, int M0, int L, int Q, double Qfloat, double threshold, int update);

/* online driver -- lwslib.h:24-26, lwslib.cpp:1424-1492 */
void TF_RTISI_LA(double *Sr, double *Si, double *wr, double *wi,
                 double *wr_asym_init, double *wi_asym_init, double *wr_asym_full, double *wi_asym_full,
                 int *w_flag, int *w_flag_ai, int *w_flag_af, double *AmpSpec,
                 int iter, int LA, int Nreal, int M, int L, int Q, double Qfloat,
                 int use_summarized_weights, double *ThresholdArray, int update);

/* not in the reference: text of the most recent HIP failure inside one of the calls above ("" if none) */
const char *lwslib_compat_last_error(void);

#endif /* LWSLIB_COMPAT_H_INCLUDED */
