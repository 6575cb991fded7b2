/* Test-only stand-in for MATLAB's mex.h / matrix.h: just enough of the (separate-complex) mx API to compile and
 * drive the gateways of matlab/*.cpp from pytest (tests/test_mex_gateways.py).  There is no MATLAB or Octave in the
 * build image or on the GPU box; this is not MATLAB's header and implements nothing beyond full double arrays. */
#ifndef LWS_TEST_MEX_H_
#define LWS_TEST_MEX_H_
#include <cstddef>

typedef size_t mwSize;
typedef struct mxArray_tag mxArray;
typedef enum { mxDOUBLE_CLASS = 6 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

bool mxIsDouble(const mxArray *a);
bool mxIsSparse(const mxArray *a);
bool mxIsComplex(const mxArray *a);
mwSize mxGetNumberOfDimensions(const mxArray *a);
const mwSize *mxGetDimensions(const mxArray *a);
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double *mxGetPr(const mxArray *a);
double *mxGetPi(const mxArray *a);
double mxGetScalar(const mxArray *a);
mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID cls, mxComplexity cplx);
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity cplx);
void mxSetM(mxArray *a, mwSize m);
void mxSetN(mxArray *a, mwSize n);
void *mxMalloc(size_t n);
void mxFree(void *p);
void mxSetData(mxArray *a, void *p);
void mxSetImagData(mxArray *a, void *p);
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);
#endif
