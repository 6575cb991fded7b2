// stream-copy variants: which launch shape reaches the HBM copy peak on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vec4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void k_copy(vec4f *__restrict__ dst, const vec4f *__restrict__ src, size_t n) {
    size_t base = ((size_t)blockIdx.x * U) * blockDim.x + threadIdx.x;
    vec4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + (size_t)u * blockDim.x; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
}
__global__ void k_gs(vec4f *__restrict__ dst, const vec4f *__restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    for (size_t bytes : {size_t(1) << 28, size_t(1) << 30, size_t(1) << 32}) {
        void *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
        size_t n = bytes / 16;
        auto rep = [&](const char *name, float ms) { printf("%-22s %5.2f GiB  %.3f ms  %.0f GB/s\n", name, bytes / 1073741824.0, ms, 2.0 * bytes / ms / 1e6); };
        rep("u1", timeit([&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(n / 256), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("u4", timeit([&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(n / 1024), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("u4 nt", timeit([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(n / 1024), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("u8", timeit([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(n / 2048), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("u8 nt", timeit([&] { hipLaunchKernelGGL((k_copy<8, true>), dim3(n / 2048), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("gs 8192", timeit([&] { hipLaunchKernelGGL(k_gs, dim3(8192), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("gs 2048", timeit([&] { hipLaunchKernelGGL(k_gs, dim3(2048), dim3(256), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("gs 1024x1024", timeit([&] { hipLaunchKernelGGL(k_gs, dim3(1024), dim3(1024), 0, 0, (vec4f *)d, (vec4f *)s, n); }));
        rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }));
        hipFree(s); hipFree(d);
    }
}
