// Test-only implementation of the mx/mex calls declared in this directory's mex.h, plus the C entry point pytest
// uses to invoke a gateway's mexFunction with numpy buffers.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mex.h"

struct mxArray_tag {
    std::vector<mwSize> dims;
    std::vector<double> re, im;      // storage the stub allocated ...
    double *xre = nullptr, *xim = nullptr;   // ... or storage handed over with mxSetData / mxSetImagData (mxMalloc'd: freed with the array)
    bool cplx = false;
    ~mxArray_tag() { free(xre); free(xim); }
};

static std::string g_log;
static void (*g_atexit)(void) = nullptr;

bool mxIsDouble(const mxArray *) { return true; }
bool mxIsSparse(const mxArray *) { return false; }
bool mxIsComplex(const mxArray *a) { return a->cplx; }
mwSize mxGetNumberOfDimensions(const mxArray *a) { return a->dims.size(); }
const mwSize *mxGetDimensions(const mxArray *a) { return a->dims.data(); }
size_t mxGetM(const mxArray *a) { return a->dims[0]; }
size_t mxGetN(const mxArray *a) {
    size_t n = 1;
    for (size_t i = 1; i < a->dims.size(); ++i) n *= a->dims[i];
    return n;
}
size_t mxGetNumberOfElements(const mxArray *a) { return a->dims[0] * mxGetN(a); }
double *mxGetPr(const mxArray *a) { return a->xre ? a->xre : const_cast<double *>(a->re.data()); }
double *mxGetPi(const mxArray *a) { return !a->cplx ? nullptr : (a->xim ? a->xim : const_cast<double *>(a->im.data())); }
double mxGetScalar(const mxArray *a) { return mxGetNumberOfElements(a) ? mxGetPr(a)[0] : 0.0; }

// the pre-2018 way of building an output (what the reference's gateways do: an empty matrix, then dimensions and mxMalloc'd planes)
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity c) {
    const mwSize dims[2] = {m, n};
    return mxCreateNumericArray(2, dims, mxDOUBLE_CLASS, c);
}
void mxSetM(mxArray *a, mwSize m) { a->dims[0] = m; }
void mxSetN(mxArray *a, mwSize n) { a->dims.resize(2); a->dims[1] = n; }
void *mxMalloc(size_t n) { return malloc(n ? n : 1); }
void mxFree(void *p) { free(p); }
void mxSetData(mxArray *a, void *p) { free(a->xre); a->xre = static_cast<double *>(p); }
void mxSetImagData(mxArray *a, void *p) { free(a->xim); a->xim = static_cast<double *>(p); a->cplx = true; }

mxArray *mxCreateNumericArray(mwSize ndim, const mwSize *dims, mxClassID, mxComplexity c) {
    mxArray *a = new mxArray_tag;
    a->dims.assign(dims, dims + ndim);
    while (a->dims.size() < 2) a->dims.push_back(1);
    a->cplx = c == mxCOMPLEX;
    a->re.assign(mxGetNumberOfElements(a), 0.0);
    if (a->cplx) a->im.assign(a->re.size(), 0.0);
    return a;
}

int mexPrintf(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_log += buf;
    return n;
}

int mexAtExit(void (*fn)(void)) {
    g_atexit = fn;
    return 0;
}

extern "C" {

// args: nrhs arrays, each (re, im-or-null, ndim, dims[3]).  Returns the number of outputs produced (0 or 1); the
// output (at most out_cap elements) goes to out_re/out_im/out_dims[3].  log receives what mexPrintf printed.
int mexstub_call(int nrhs, const double **re, const double **im, const int *ndim, const long *dims, double *out_re,
                 double *out_im, long *out_dims, long out_cap, char *log, int log_cap) {
    g_log.clear();
    std::vector<mxArray_tag> in(nrhs);
    std::vector<const mxArray *> prhs(nrhs);
    for (int i = 0; i < nrhs; ++i) {
        size_t n = 1;
        for (int d = 0; d < ndim[i]; ++d) {
            in[i].dims.push_back((mwSize)dims[3 * i + d]);
            n *= (size_t)dims[3 * i + d];
        }
        in[i].re.assign(re[i], re[i] + n);
        in[i].cplx = im[i] != nullptr;
        if (im[i]) in[i].im.assign(im[i], im[i] + n);
        prhs[i] = &in[i];
    }
    mxArray *plhs[1] = {nullptr};
    mexFunction(1, plhs, nrhs, prhs.data());
    int produced = 0;
    if (plhs[0]) {
        const long n = (long)mxGetNumberOfElements(plhs[0]);
        if (n <= out_cap) {
            memcpy(out_re, mxGetPr(plhs[0]), n * sizeof(double));
            if (plhs[0]->cplx) memcpy(out_im, mxGetPi(plhs[0]), n * sizeof(double));
            for (int d = 0; d < 3; ++d) out_dims[d] = d < (int)plhs[0]->dims.size() ? (long)plhs[0]->dims[d] : 1;
            produced = 1;
        }
        delete plhs[0];
    }
    snprintf(log, log_cap, "%s", g_log.c_str());
    return produced;
}

void mexstub_exit(void) {
    if (g_atexit) g_atexit();
    g_atexit = nullptr;
}
}
