#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int NACC, int MODE>
__global__ void k(float *out, int iters, float a, float b) {
    v2f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    v2f av = {a, a * 1.0001f}, bv = {b, b * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (MODE == 0) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);          // pk_fma
                else if (MODE == 1) { acc[i].x = fmaf(acc[i].x, a, b); acc[i].y = fmaf(acc[i].y, a, b); asm volatile("" : "+v"(acc[i].x), "+v"(acc[i].y)); }  // 2 scalar fma
                else if (MODE == 2) acc[i] = acc[i] + av;                                    // pk_add
                else { acc[i] = __builtin_elementwise_fma(acc[i].yx, av, bv); }             // pk_fma with swizzle
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MODE> void run(int threads, const char *name) {
    float *d; hipMalloc(&d, 1 << 23);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, MODE><<<256, threads>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<NACC, MODE><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16 * NACC;   // float2 operations per wave
    int wps = threads / 256;
    printf("%s NACC=%d threads=%d: %.3f ms -> %.2f ns per float2-op per SIMD (%d waves/SIMD) = %.2f clk @2.4GHz\n", name, NACC, threads, ms, ms * 1e6 / (n * wps), wps, ms * 1e6 / (n * wps) * 2.4);
    hipFree(d);
}
int main() {
    run<8, 0>(512, "pk_fma      "); run<8, 1>(512, "2x scalar   "); run<8, 2>(512, "pk_add      "); run<8, 3>(512, "pk_fma swz  ");
    run<8, 0>(256, "pk_fma      "); run<8, 1>(256, "2x scalar   ");
    run<2, 0>(512, "pk_fma 2acc "); run<2, 1>(512, "2x scalar 2acc");
    return 0;
}
