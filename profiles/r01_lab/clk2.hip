#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void ksleep(long long *out, int iters) {
    long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
}
// 256 dependent fmas per iteration, fully unrolled: loop overhead negligible
__global__ void kchain(long long *out, int iters, float m, float b) {
    float a = threadIdx.x * 0.001f;
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 256; ++u) a = fmaf(a, m, b);
    }
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = (long long)a; }
}
int main() {
    long long *d, h[2]; hipMalloc(&d, 64);
    for (int blocks : {1, 256}) {
        ksleep<<<blocks, 64>>>(d, 20000); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double sec = h[0] / 1e8;
        printf("sleep: blocks %d: %.4f s for %d x s_sleep(127) (= %d x 8128 cycles if 64 clk units) -> %.3f GHz; clock64/wall = %.2f\n", blocks, sec, 20000, 20000, 20000.0 * 8128 / sec / 1e9, (double)h[1] / h[0]);
    }
    for (int blocks : {1, 256, 1024}) for (int thr : {64, 256, 512}) {
        kchain<<<blocks, thr>>>(d, 20000, 1.0001f, 0.5f); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double sec = h[0] / 1e8;
        printf("chain: blocks %d thr %d: %.4f s for %.0f dependent fmas per wave -> %.2f ns each\n", blocks, thr, sec, 20000.0 * 256, sec / (20000.0 * 256) * 1e9);
    }
}
