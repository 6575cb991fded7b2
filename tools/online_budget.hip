// online_budget.hip -- per-phase clock budget of the online LDS kernel (lws_amd/csrc/lws_online.hip: k_online4) on BASELINE config 3's
// shape, without Python: random state / magnitudes / weights (timing does not depend on the values), launch_online_lds directly, HIP
// events around it, and -- built with -DLWS_LAB=1 or 2 -- the s_memtime stamps every wave of workgroup 0 sums up per phase
// (lws_online.hip: LAB / LAB2).  Prints one JSON object; tools/online_budget.sh builds the three variants (no stamps / level 1 /
// level 2), runs them and writes profiles/r05_online_phase_budget.json.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-sched-strategy=max-ilp [-DLWS_LAB=1|2] \
//         -I include -I lws_amd/csrc tools/online_budget.hip -o tools/online_budget_l<n>
//   ./online_budget_l<n> [B] [T] [F] [LA] [iters] [reps]
#include "../lws_amd/csrc/lws_online.hip"
#include <cstdio>
#include <vector>
#include <random>

namespace lws { int set_error(int code, const char *, ...) { return code; } }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 500, F = argc > 3 ? atoi(argv[3]) : 513;
    const int LA = argc > 4 ? atoi(argv[4]) : 3, iters = argc > 5 ? atoi(argv[5]) : 10, reps = argc > 6 ? atoi(argv[6]) : 3;
    const int Q = 4, L = 5, K1 = L + 1, Np = F + 2 * L, Tp = T + 2 * (Q - 1);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t n = (size_t)B * Tp * Np;
    std::vector<float2> hs(n);
    std::vector<float> ha(n);
    for (size_t i = 0; i < n; ++i) { hs[i] = make_float2(nd(rng), nd(rng)); ha[i] = sqrtf(hs[i].x * hs[i].x + hs[i].y * hs[i].y); }
    std::vector<float2> hw(3 * Q * Q * K1);
    for (int s = 0; s < 3; ++s)
        for (int r = 0; r < Q; ++r)
            for (int k = 0; k < K1; ++k) {
                const float2 b = make_float2(nd(rng) * 0.1f, nd(rng) * 0.1f);
                for (int p = 0; p < Q; ++p) {
                    const double ang = 2.0 * M_PI * p * r / Q;
                    hw[((s * Q + p) * Q + r) * K1 + k] = make_float2((float)(b.x * cos(ang) - b.y * sin(ang)), (float)(b.x * sin(ang) + b.y * cos(ang)));
                }
            }
    std::vector<float> hthr((size_t)B * iters, 0.f);
    float2 *ds, *ds0, *dw; float *da, *dthr;
    CK(hipMalloc(&ds, n * 8)); CK(hipMalloc(&ds0, n * 8)); CK(hipMalloc(&da, n * 4)); CK(hipMalloc(&dw, hw.size() * 8)); CK(hipMalloc(&dthr, hthr.size() * 4));
    CK(hipMemcpy(ds0, hs.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dthr, hthr.data(), hthr.size() * 4, hipMemcpyHostToDevice));
    lws::GenericArgs<float> g{};
    g.state = ds; g.amp = da; g.thr = dthr;
    for (int s = 0; s < 3; ++s) { g.w[s].w = dw + s * Q * Q * K1; g.w[s].flag = nullptr; }
    g.F = F; g.T = T; g.L = L; g.Q = Q; g.Qp = Q; g.n_thr = iters; g.LA = LA; g.update = 2; g.qdiv = 4.f; g.mode = lws::MODE_ONLINE;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemcpy(ds, ds0, n * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(e0, 0));
        CK(lws::launch_online_lds(g, B, Q, 1, nullptr, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<float2> out((size_t)Tp * Np);
    CK(hipMemcpy(out.data(), ds, out.size() * 8, hipMemcpyDeviceToHost));
    double cs = 0; for (auto &v : out) cs += fabs(v.x) * 1.25 + fabs(v.y);
    int lab_level = 0;
#ifdef LWS_LAB
    lab_level = LWS_LAB;
#endif
    printf("{\"B\": %d, \"T\": %d, \"F\": %d, \"LA\": %d, \"iters\": %d, \"lab_level\": %d, \"kernel_ms\": %.3f, \"checksum_spectrogram0\": %.9e",
           B, T, F, LA, iters, lab_level, best, cs);
#ifdef LWS_LAB
    std::vector<unsigned long long> lab(LAB_N);
    CK(hipMemcpyFromSymbol(lab.data(), HIP_SYMBOL(lws::g_lab), LAB_N * 8));
    const long long steps = lab[0] ? (long long)lab[0] : 1;
    printf(", \"steps\": %lld, \"waves\": {", steps);
    bool first = true;
    for (int w = 0; w < 16; ++w) {
        const unsigned long long *p = lab.data() + 8 + w * 8;
        if (!p[0] && !p[1] && !p[2] && !p[3]) continue;
        printf("%s\"hw%d\": [%.1f, %.1f, %.1f, %.1f]", first ? "" : ", ", w, (double)p[0] / steps, (double)p[1] / steps, (double)p[2] / steps, (double)p[3] / steps);
        first = false;
    }
    printf("}");
#endif
    printf("}\n");
    return 0;
}
