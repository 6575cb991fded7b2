#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ void k(float *out, int iters, float a, float b) {
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 0.001f + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}
template <int NACC> void run(int threads, const char *name) {
    float *d; hipMalloc(&d, (1 << 23) + 64);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<NACC><<<256, threads>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, ((long long *)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    double ninstr_per_wave = (double)iters * 16 * NACC;
    int waves_per_simd = threads / 256;
    printf("%s NACC=%d threads=%d: %.3f ms, clock64 delta %lld, per-wave instr %.0f -> %.2f clock64-ticks/instr/wave; wall: %.2f ns per instr per SIMD (%d waves/SIMD)\n",
           name, NACC, threads, ms, cyc, ninstr_per_wave, cyc / ninstr_per_wave, ms * 1e6 / (ninstr_per_wave * (waves_per_simd ? waves_per_simd : 1)), waves_per_simd);
    hipFree(d);
}
int main() {
    run<1>(256, "dep-chain"); run<2>(256, "2acc"); run<4>(256, "4acc"); run<8>(256, "8acc");
    run<1>(512, "dep-chain"); run<2>(512, "2acc"); run<4>(512, "4acc"); run<8>(512, "8acc");
    run<8>(1024, "8acc"); run<2>(1024, "2acc");
    return 0;
}
