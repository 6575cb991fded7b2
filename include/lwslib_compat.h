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

/* Argument groups shared by the entry points (types and order are the reference's; the names are ours):
 *   planes   real and imaginary plane of the extended spectrogram, row pitch bins + 2*half_width, updated in place
 *   weights  real plane, imaginary plane and participation flags of one weight tensor [Qp][Q][half_width + 1]
 *   target   magnitudes the updated bins are projected to (same layout as the planes) */
#define LWSLIB_PLANES double *re, double *im
#define LWSLIB_WEIGHTS double *w_re, double *w_im, int *w_on
#define LWSLIB_SWEEP LWSLIB_PLANES, LWSLIB_WEIGHTS, double *target

/* helpers -- lwslib.h:6-8, lwslib.cpp:15-65: pad to / cut from the extended layout, magnitudes */
void ExtendSpec(double *ext_re, double *ext_im, double *in_re, double *in_im, int bins, int frames, int half_width, int overlap);
void CopySpec(double *ext_re, double *ext_im, double *out_re, double *out_im, int bins, int frames, int half_width, int overlap);
void ComputeAmpSpec(LWSLIB_PLANES, double *magnitude, int count);

/* batch sweeps -- lwslib.h:10-13, lwslib.cpp:72-467 (level: bins with target <= level are left alone) */
void LWSQ2(LWSLIB_SWEEP, int bins, int frames, int half_width, double level);
void LWSQ4(LWSLIB_SWEEP, int bins, int frames, int half_width, double level);
void LWSanyQ(LWSLIB_SWEEP, int bins, int frames, int half_width, int overlap, double level);
void LWSfractionalQ(LWSLIB_SWEEP, int bins, int frames, int half_width, int overlap, double level);

/* sweeps that use past frames only -- lwslib.h:15-18, lwslib.cpp:473-764.
 * NoFuture_LWSQ4 reproduces the reference's addressing (flat offset (m-r)*Np + 2n +- k, lwslib.cpp:559-594). */
void NoFuture_LWSQ2(LWSLIB_SWEEP, int bins, int frames, int half_width, double level);
void NoFuture_LWSQ4(LWSLIB_SWEEP, int bins, int frames, int half_width, double level);
void NoFuture_LWSanyQ(LWSLIB_SWEEP, int bins, int frames, int half_width, int overlap, double level);
void NoFuture_LWSfractionalQ(LWSLIB_SWEEP, int bins, int frames, int half_width, int overlap, double level);

/* sweeps over `frames` frames that may read `usable_right` frames to their right -- lwslib.h:20-23,
 * lwslib.cpp:776-1421 (self_term 1: add S/Q to the centre sum; the shipped callers pass 2) */
void Asym_UpdatePhaseQ2(LWSLIB_SWEEP, int bins, int frames, int usable_right, int half_width, double level, int self_term);
void Asym_UpdatePhaseQ4(LWSLIB_SWEEP, int bins, int frames, int usable_right, int half_width, double level, int self_term);
void Asym_UpdatePhaseanyQ(LWSLIB_SWEEP, int bins, int frames, int usable_right, int half_width, int overlap, double level,
                          int self_term);
void Asym_UpdatePhasefractionalQ(LWSLIB_SWEEP, int bins, int frames, int usable_right, int half_width, int overlap,
                                 double overlap_exact, double level, int self_term);

/* online driver -- lwslib.h:24-26, lwslib.cpp:1424-1492: the weights of the symmetric window, of the envelope used for a
 * frame's first estimate and of the envelope of the newest frame; one level per iteration */
void TF_RTISI_LA(LWSLIB_PLANES, double *w_re, double *w_im, double *first_re, double *first_im, double *newest_re,
                 double *newest_im, int *w_on, int *first_on, int *newest_on, double *target, int iterations,
                 int look_ahead, int bins, int frames, int half_width, int overlap, double overlap_exact,
                 int summarised_weights, double *levels, int self_term);

#undef LWSLIB_SWEEP
#undef LWSLIB_WEIGHTS
#undef LWSLIB_PLANES

/* not in the reference: text of the most recent HIP failure inside one of the calls above ("" if none) */
const char *lwslib_compat_last_error(void);

#endif /* LWSLIB_COMPAT_H_INCLUDED */
