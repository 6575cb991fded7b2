// lws_stft.hip -- the steps either side of the LWS path, on the device: STFT, inverse STFT and the consistency measure
// 20 log10(|S| / |STFT(iSTFT(S)) - S|) of lws.pyx:43-144, for batches of independent signals / spectrograms.
//
// One workgroup per frame transforms it in LDS, fp32, for any even frame size N in [32, 4096] (lws.pyx:43-90 accepts any even
// size).  N = m 2^a with m odd: the m interleaved subsequences of length 2^a go through a radix-2 Stockham FFT (all of them in
// every butterfly stage), then one stage of m-point DFTs with the twiddles exp(-2 pi j r k / N) combines them (Cooley-Tukey,
// decimation in time): N (a + m) operations -- N log2 N for a power of two, 2.6 N log2 N for 1536 = 3 x 512, N^2 / 8 for
// 1000 = 125 x 8 (round 2 ran a direct N^2 DFT for every size that is not a power of two);
// the overlap-add is a gather (each output sample sums the <= ceil(N/hop) frames that cover it), so there are no
// atomics and the result does not depend on scheduling.  Sums of squares for the consistency are accumulated in fp64
// per frame and reduced in a fixed order.
#include "../../include/lws_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <vector>

#include "lws_common.h"

namespace {

constexpr int MAXN = 4096, MINN = 32, FFT_THREADS = 256;   // two N-point complex buffers (three if N is not a power of two): <= 96 KB of LDS

#define STFT_TRY(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            return lws::set_error(LWS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Complex DFT of n = m 2^a points held in LDS (x: data, y: scratch of the same size, followed -- if m > 1 -- by n entries for
// the twiddle table), by all threads of the block; natural order in and out.  sign = -1 forward, +1 inverse (unnormalised).
// Returns the buffer that holds the result.
__device__ float2 *fft_lds(float2 *x, float2 *y, int n, int m, int a, float sign) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n2 = n / m;                                // 2^a
    float2 *tw = y + n;
    if (m > 1) {
        // exp(sign 2 pi j i / n), and the m interleaved subsequences side by side: y[r n2 + j] = x[m j + r]
        for (int i = tid; i < n; i += nthr) {
            float sn, cs;
            sincospif(2.0f * (float)i / (float)n, &sn, &cs);
            tw[i] = make_float2(cs, sign * sn);
            const int j = i / m, r = i - j * m;
            y[r * n2 + j] = x[i];
        }
        __syncthreads();
        float2 *t = x; x = y; y = t;
    }
    // radix-2 Stockham (auto-sort, decimation in frequency) of the m blocks of n2 points, all blocks in every stage
    int ncur = n2, s = 1;
    for (int st = 0; st < a; ++st) {
        const int h = ncur >> 1;
        for (int i0 = tid; i0 < n / 2; i0 += nthr) {
            const int blk = i0 / (n2 / 2), i = i0 - blk * (n2 / 2);
            const int p = i / s, q = i - p * s;          // s is a power of two: shifts
            float sn, cs;
            sincospif(sign * 2.0f * (float)p / (float)ncur, &sn, &cs);
            const float2 *xb = x + blk * n2;
            float2 *yb = y + blk * n2;
            const float2 u = xb[q + s * p], v = xb[q + s * (p + h)];
            const float2 d = make_float2(u.x - v.x, u.y - v.y);
            yb[q + s * (2 * p)] = make_float2(u.x + v.x, u.y + v.y);
            yb[q + s * (2 * p + 1)] = make_float2(d.x * cs - d.y * sn, d.x * sn + d.y * cs);
        }
        __syncthreads();
        float2 *t = x; x = y; y = t;
        ncur = h;
        s <<= 1;
    }
    if (m == 1) return x;
    // X[k + n2 q] = sum_r exp(sign 2 pi j r (k + n2 q) / n) Y_r[k]
    for (int o = tid; o < n; o += nthr) {
        const int k = o % n2;
        float ar = 0.f, ai = 0.f;
        int idx = 0;                                     // (r o) mod n
        for (int r = 0; r < m; ++r) {
            const float2 v = x[r * n2 + k], w = tw[idx];
            ar += v.x * w.x - v.y * w.y;
            ai += v.x * w.y + v.y * w.x;
            idx += o;
            if (idx >= n) idx -= n;
        }
        y[o] = make_float2(ar, ai);
    }
    __syncthreads();
    return y;
}

// lws.pyx:118-128: one frame = inverse FFT of the Hermitian completion, first N samples, times the synthesis window
__global__ void __launch_bounds__(FFT_THREADS) k_istft_frames(const float2 *S, float *frames, const float *swin, int M,
                                                               int N, int odd, int log2e) {
    extern __shared__ float2 lds[];
    const int m = blockIdx.x, b = blockIdx.y, F = N / 2 + 1;
    const float2 *row = S + ((size_t)b * M + m) * F;
    float2 *x = lds, *y = lds + N;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        float2 v = row[k < F ? k : N - k];
        if (k >= F) v.y = -v.y;
        x[k] = v;
    }
    __syncthreads();
    const float2 *r = fft_lds(x, y, N, odd, log2e, 1.0f);
    const float inv = 1.0f / (float)N;
    float *out = frames + ((size_t)b * M + m) * N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) out[n] = r[n].x * inv * swin[n];
}

// lws.pyx:123-128: signal[t] = sum over the frames that cover t; samples t < zero_lo or t >= Tfull - zero_hi are
// written as zero (what cutting them off and padding zeros back does, lws.pyx:55-67 after 130-137).
__global__ void k_overlap_add(const float *frames, float *signal, int M, int N, int hop, int Tfull, int zero_lo, int zero_hi) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tfull) return;
    float acc = 0.f;
    if (t >= zero_lo && t < Tfull - zero_hi) {
        int s_hi = t / hop;
        if (s_hi > M - 1) s_hi = M - 1;
        int s_lo = (t - N + hop) / hop;      // smallest s with t - s*hop < N  (ceil((t - N + 1) / hop))
        if (t - N + 1 <= 0) s_lo = 0;
        const float *f = frames + (size_t)b * M * N;
        for (int s = s_lo; s <= s_hi; ++s) acc += f[(size_t)s * N + (t - s * hop)];   // ascending s, as the reference adds them
    }
    signal[(size_t)b * Tfull + t] = acc;
}

// lws.pyx:82-88: frame m = x[m*hop + n - pre] * awin[n] (zero outside the signal; zero for n >= fs when the transform is longer
// than the frame: np.fft.fft(frame, n = fftsize) pads at the end), FFT, bins 0..N/2.
// S_out != null: write the spectrogram.  rows != null: accumulate |X - S_ref|^2 and |S_ref|^2 of the frame in fp64.
__global__ void __launch_bounds__(FFT_THREADS) k_stft_frames(const float *x, int len, int pitch, int pre, const float *awin,
                                                              float2 *S_out, const float2 *S_ref, double *rows, int M,
                                                              int fs, int N, int odd, int log2e, int hop) {
    extern __shared__ float2 lds[];
    __shared__ double red[2][FFT_THREADS];
    const int m = blockIdx.x, b = blockIdx.y, F = N / 2 + 1;
    float2 *xa = lds, *ya = lds + N;
    const float *sig = x + (size_t)b * pitch;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int i = m * hop + n - pre;
        xa[n] = make_float2((n < fs && i >= 0 && i < len) ? sig[i] * awin[n] : 0.f, 0.f);
    }
    __syncthreads();
    const float2 *r = fft_lds(xa, ya, N, odd, log2e, -1.0f);
    if (S_out) {
        float2 *o = S_out + ((size_t)b * M + m) * F;
        for (int k = threadIdx.x; k < F; k += blockDim.x) o[k] = r[k];
    }
    if (rows) {
        const float2 *ref = S_ref + ((size_t)b * M + m) * F;
        double e = 0, p = 0;
        for (int k = threadIdx.x; k < F; k += blockDim.x) {
            const float2 s = ref[k];
            const double dx = (double)r[k].x - s.x, dy = (double)r[k].y - s.y;
            e += dx * dx + dy * dy;
            p += (double)s.x * s.x + (double)s.y * s.y;
        }
        red[0][threadIdx.x] = e;
        red[1][threadIdx.x] = p;
        __syncthreads();
        for (int s2 = blockDim.x / 2; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) { red[0][threadIdx.x] += red[0][threadIdx.x + s2]; red[1][threadIdx.x] += red[1][threadIdx.x + s2]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { rows[((size_t)b * M + m) * 2] = red[1][0]; rows[((size_t)b * M + m) * 2 + 1] = red[0][0]; }
    }
}

__global__ void k_sum_rows(const double *rows, double *out, int M, int B) {   // one thread per spectrogram, fixed order
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double p = 0, e = 0;
    for (int m = 0; m < M; ++m) { p += rows[((size_t)b * M + m) * 2]; e += rows[((size_t)b * M + m) * 2 + 1]; }
    out[2 * b] = p;
    out[2 * b + 1] = e;
}

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return LWS_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return lws::set_error(LWS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return LWS_OK;
    }
};
// Windows and scratch of the transforms, one set per device, shared by every caller.  The entry points are asynchronous
// on the caller's stream, so a context remembers (event) the last work enqueued with it: the next call -- possibly on
// another stream, with another window or a larger shape -- first makes its own stream wait for that event, i.e. users of
// one device's context are serialised ON THE DEVICE (never on the host); a buffer is only re-allocated after hipFree,
// which waits for the device.  The host mutex covers the bookkeeping and the enqueue order.
struct DeviceCtx {
    Scratch frames, signal, rows, out, win_a, win_s;
    hipEvent_t last = nullptr;
    bool busy = false;
};
constexpr int MAX_DEVICES = 64;
std::mutex g_mu;
DeviceCtx g_ctx[MAX_DEVICES];

int ctx_enter(DeviceCtx &c, hipStream_t s) {
    if (!c.last) STFT_TRY(hipEventCreateWithFlags(&c.last, hipEventDisableTiming));
    if (c.busy) STFT_TRY(hipStreamWaitEvent(s, c.last, 0));
    return LWS_OK;
}
int ctx_leave(DeviceCtx &c, hipStream_t s) {
    STFT_TRY(hipEventRecord(c.last, s));
    c.busy = true;
    return LWS_OK;
}

// N = odd * 2^log2e
struct Factors { int odd, log2e; };
Factors factor(int n) { Factors f{n, 0}; while (!(f.odd & 1)) { f.odd >>= 1; ++f.log2e; } return f; }
size_t fft_lds_bytes(int N) { return (size_t)(factor(N).odd > 1 ? 3 : 2) * N * sizeof(float2); }
// frames of more than 2048 points need more dynamic LDS than a kernel gets by default
template <typename K> hipError_t allow_lds(K kernel) {
    static std::atomic<unsigned long long> done{0};   // one bit per device
    int dev;
    if (!lws::attr_needed(done, &dev)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * MAXN * (int)sizeof(float2));
    if (e == hipSuccess) lws::attr_done(done, dev);
    return e;
}

int check_shape(int device, int B, int M, int N, int hop) {
    if (device < 0 || device >= MAX_DEVICES) return lws::set_error(LWS_ERR_INVALID, "device index %d out of range", device);
    if (B < 0 || M < 1) return lws::set_error(LWS_ERR_INVALID, "empty batch or no frames");
    if (N < MINN || N > MAXN || (N & 1)) return lws::set_error(LWS_ERR_UNSUPPORTED, "frame size %d: the device transform serves even sizes in [%d, %d]", N, MINN, MAXN);
    if (hop < 1 || hop > N) return lws::set_error(LWS_ERR_INVALID, "frame shift %d", hop);
    return LWS_OK;
}
int allow_lds_all() {
    STFT_TRY(allow_lds(k_stft_frames));
    STFT_TRY(allow_lds(k_istft_frames));
    return LWS_OK;
}

int upload_window(Scratch &dst, const double *w, int N, hipStream_t s) {
    if (!w) return lws::set_error(LWS_ERR_INVALID, "null window");
    std::vector<float> f(N);
    for (int i = 0; i < N; ++i) f[i] = (float)w[i];
    int rc = dst.ensure((size_t)N * sizeof(float));
    if (rc) return rc;
    STFT_TRY(hipMemcpyAsync(dst.p, f.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice, s));
    STFT_TRY(hipStreamSynchronize(s));   // f goes out of scope
    return LWS_OK;
}

int prepad(int N, int hop) { const int r = N % hop; return r == 0 ? N - hop : N - r; }   // lws.pyx:55-60

}  // namespace

extern "C" {

int lws_stft_frames(int len, int N, int fshift, int perfectrec) {
    if (N < 1 || fshift < 1 || len < 0) return -1;
    if (perfectrec) {
        const long total = (long)prepad(N, fshift) + len + ((fshift - len % fshift) % fshift);
        return (int)(total / fshift);
    }
    const long r = ((long)len - N) % fshift;
    const long padded = len + ((fshift - (r < 0 ? r + fshift : r)) % fshift);
    return (int)((padded - N) / fshift + 1);
}

int lws_istft_length(int M, int N, int fshift, int perfectrec) {
    const int Tfull = fshift * (M - 1) + N;
    return perfectrec ? Tfull - prepad(N, fshift) - (N - fshift) : Tfull;
}

// fs: samples per frame (the window's length), N >= fs: points of the transform (fs..N-1 are zeros)
static int stft_impl(int device, const float *x_dev, int B, int len, int fs, int N, int fshift, const double *awin, int perfectrec,
                     void *S_dev, void *stream) {
    const int M = lws_stft_frames(len, fs, fshift, perfectrec);
    int rc = check_shape(device, B, M, N, fshift);
    if (rc) return rc;
    if (fs < 2 || fs > N || (fs & 1) || fshift > fs) return lws::set_error(LWS_ERR_INVALID, "frame of %d samples, transform of %d points, shift %d", fs, N, fshift);
    if (!x_dev || !S_dev) return lws::set_error(LWS_ERR_INVALID, "null device pointer");
    if (B == 0) return LWS_OK;
    STFT_TRY(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_ctx[device];
    if ((rc = ctx_enter(c, s))) return rc;
    if ((rc = allow_lds_all())) return rc;
    if ((rc = upload_window(c.win_a, awin, fs, s))) return rc;
    hipLaunchKernelGGL(k_stft_frames, dim3(M, B), dim3(FFT_THREADS), fft_lds_bytes(N), s, x_dev, len, len,
                       perfectrec ? prepad(fs, fshift) : 0, static_cast<const float *>(c.win_a.p),
                       static_cast<float2 *>(S_dev), nullptr, nullptr, M, fs, N, factor(N).odd, factor(N).log2e, fshift);
    STFT_TRY(hipGetLastError());
    return ctx_leave(c, s);
}

int lws_istft_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *swin, int perfectrec,
                  float *x_dev, void *stream) {
    int rc = check_shape(device, B, M, N, fshift);
    if (rc) return rc;
    if (!x_dev || !S_dev) return lws::set_error(LWS_ERR_INVALID, "null device pointer");
    if (B == 0) return LWS_OK;
    STFT_TRY(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_ctx[device];
    if ((rc = ctx_enter(c, s))) return rc;
    if ((rc = allow_lds_all())) return rc;
    if ((rc = upload_window(c.win_s, swin, N, s))) return rc;
    const int Tfull = fshift * (M - 1) + N, out_len = lws_istft_length(M, N, fshift, perfectrec);
    if ((rc = c.frames.ensure((size_t)B * M * N * sizeof(float)))) return rc;
    if ((rc = c.signal.ensure((size_t)B * Tfull * sizeof(float)))) return rc;
    hipLaunchKernelGGL(k_istft_frames, dim3(M, B), dim3(FFT_THREADS), fft_lds_bytes(N), s,
                       static_cast<const float2 *>(S_dev), static_cast<float *>(c.frames.p),
                       static_cast<const float *>(c.win_s.p), M, N, factor(N).odd, factor(N).log2e);
    hipLaunchKernelGGL(k_overlap_add, dim3((Tfull + 255) / 256, B), dim3(256), 0, s, static_cast<const float *>(c.frames.p),
                       static_cast<float *>(c.signal.p), M, N, fshift, Tfull, 0, 0);
    STFT_TRY(hipGetLastError());
    // lws.pyx:130-137: cut the leading pad and the last N - hop samples
    const int off = perfectrec ? prepad(N, fshift) : 0;
    STFT_TRY(hipMemcpy2DAsync(x_dev, (size_t)out_len * sizeof(float), static_cast<const float *>(c.signal.p) + off,
                              (size_t)Tfull * sizeof(float), (size_t)out_len * sizeof(float), B, hipMemcpyDeviceToDevice, s));
    return ctx_leave(c, s);
}

int lws_stft_dev(int device, const float *x_dev, int B, int len, int N, int fshift, const double *awin, int perfectrec,
                 void *S_dev, void *stream) {
    return stft_impl(device, x_dev, B, len, N, N, fshift, awin, perfectrec, S_dev, stream);
}

int lws_stft_zp_dev(int device, const float *x_dev, int B, int len, int fsize, int fftsize, int fshift, const double *awin,
                    int perfectrec, void *S_dev, void *stream) {
    return stft_impl(device, x_dev, B, len, fsize, fftsize, fshift, awin, perfectrec, S_dev, stream);
}

int lws_consistency_dev(int device, const void *S_dev, int B, int M, int N, int fshift, const double *awin,
                        const double *swin, int perfectrec, double *out, void *stream) {
    int rc = check_shape(device, B, M, N, fshift);
    if (rc) return rc;
    if (!S_dev || !out) return lws::set_error(LWS_ERR_INVALID, "null pointer");
    if (B == 0) return LWS_OK;
    STFT_TRY(hipSetDevice(device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_ctx[device];
    if ((rc = ctx_enter(c, s))) return rc;
    if ((rc = allow_lds_all())) return rc;
    if ((rc = upload_window(c.win_a, awin, N, s))) return rc;
    if ((rc = upload_window(c.win_s, swin, N, s))) return rc;
    const int Tfull = fshift * (M - 1) + N;
    if ((rc = c.frames.ensure((size_t)B * M * N * sizeof(float)))) return rc;
    if ((rc = c.signal.ensure((size_t)B * Tfull * sizeof(float)))) return rc;
    if ((rc = c.rows.ensure((size_t)B * M * 2 * sizeof(double)))) return rc;
    if ((rc = c.out.ensure((size_t)B * 2 * sizeof(double)))) return rc;
    const float2 *S = static_cast<const float2 *>(S_dev);
    hipLaunchKernelGGL(k_istft_frames, dim3(M, B), dim3(FFT_THREADS), fft_lds_bytes(N), s, S,
                       static_cast<float *>(c.frames.p), static_cast<const float *>(c.win_s.p), M, N, factor(N).odd, factor(N).log2e);
    // with perfectrec the reference cuts the first prepad and the last N - hop samples and the forward transform pads
    // zeros back in their place (same frame count): the full overlap-add signal with those samples zeroed
    hipLaunchKernelGGL(k_overlap_add, dim3((Tfull + 255) / 256, B), dim3(256), 0, s, static_cast<const float *>(c.frames.p),
                       static_cast<float *>(c.signal.p), M, N, fshift, Tfull, perfectrec ? prepad(N, fshift) : 0,
                       perfectrec ? N - fshift : 0);
    hipLaunchKernelGGL(k_stft_frames, dim3(M, B), dim3(FFT_THREADS), fft_lds_bytes(N), s,
                       static_cast<const float *>(c.signal.p), Tfull, Tfull, 0, static_cast<const float *>(c.win_a.p),
                       static_cast<float2 *>(nullptr), S, static_cast<double *>(c.rows.p), M, N, N, factor(N).odd, factor(N).log2e, fshift);
    hipLaunchKernelGGL(k_sum_rows, dim3((B + 63) / 64), dim3(64), 0, s, static_cast<const double *>(c.rows.p),
                       static_cast<double *>(c.out.p), M, B);
    STFT_TRY(hipGetLastError());
    STFT_TRY(hipMemcpyAsync(out, c.out.p, (size_t)B * 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    STFT_TRY(hipStreamSynchronize(s));
    return ctx_leave(c, s);
}

}  // extern "C"
