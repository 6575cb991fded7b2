#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long *out, int iters) {
    float a = threadIdx.x * 0.001f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 0.5f);
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
int main() {
    long long *d, h[3]; hipMalloc(&d, 64);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int sclk = 0; hipDeviceGetAttribute(&sclk, hipDeviceAttributeClockRate, 0);
    for (int blocks : {1, 256, 2048}) for (int thr : {64, 512}) {
        k<<<blocks, thr>>>(d, 4000000); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %d thr %d: clock64 %lld wall %lld (wall rate %d kHz, attr sclk %d kHz) -> shader clock %.3f GHz; %.2f clk per dependent fma\n", blocks, thr, h[0], h[1], rate, sclk, (double)h[0] / ((double)h[1] / rate) / 1e6, (double)h[0] / 4e6);
    }
}
