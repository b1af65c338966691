// ubench_pair.hip -- the pair recurrence kernels alone (diagnostic, not shipped): global-K vs LDS-fed, and where the
// LDS-fed kernel's extra time goes (helper without loads / without any work).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sparsernns_amd/csrc tools/ubench_pair.hip -o tools/bin/ubench_pair
#include "scan_quad.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace s5;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <class F>
double time_us(F launch, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 32, P = 64, L = argc > 2 ? atoi(argv[2]) : 4096, TB = L / 4, PG = P / 32;
    const int ea = 15, k_re = 65536 - 2;
    const size_t runs = (size_t)B * PG, kwords = runs * TB * 256, halves = runs * TB * 256;
    std::vector<int32_t> K(kwords + 64 * 256), ar(P), ai(P);
    std::vector<int16_t> b16(halves + 64 * 256);
    srand(7);
    // lane l of a run, block tb, slot s in [t0 t2 t1 t3]: Bu value v; K = (v << 16) + k(role)
    for (size_t run = 0; run < runs; ++run)
        for (int tb = 0; tb < TB; ++tb)
            for (int l = 0; l < 64; ++l)
                for (int s = 0; s < 4; ++s) {
                    const int v = (rand() % 4001) - 2000;
                    const bool odd_step = s >= 2, laneB = l & 1;
                    const bool re_role = laneB ? !odd_step : odd_step; // even steps: lane B computes re'
                    K[((run * TB + tb) * 64 + l) * 4 + s] = (int32_t)((uint32_t)v << 16) + (re_role ? k_re : 0);
                    b16[((run * (TB / 2) + tb / 2) * 64 + l) * 8 + 4 * (tb & 1) + s] = (int16_t)v;
                }
    for (int p = 0; p < P; ++p) { ar[p] = 32000 - 41 * p; ai[p] = (p % 2 ? 1 : -1) * (900 + 97 * p); }
    int32_t *dK, *dar, *dai; int16_t *db16, *dxs1, *dxs2;
    CK(hipMalloc(&dK, K.size() * 4)); CK(hipMalloc(&db16, b16.size() * 2)); CK(hipMalloc(&dar, P * 4)); CK(hipMalloc(&dai, P * 4));
    CK(hipMalloc(&dxs1, halves * 2)); CK(hipMalloc(&dxs2, halves * 2));
    CK(hipMemcpy(dK, K.data(), K.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db16, b16.data(), b16.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dar, ar.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dai, ai.data(), P * 4, hipMemcpyHostToDevice));
    ScanPairArgs g{dK, dxs1, dar, dai, B, TB, P, ea, ea};
    ScanPairLArgs l{db16, dxs2, dar, dai, B, TB, P, ea, ea};
    const double algo = (double)B * L * P * 16;
    auto report = [&](const char *name, double us) { printf("%-44s %8.2f us  %6.2f ns/step  %.3f of 8 TB/s (16*P B/frame)\n", name, us, us * 1e3 / L, algo / (us * 1e-6) / 8e12); };
    report("pair, K int32 from global", time_us([&] { hipLaunchKernelGGL(k_scan_pair_asm, dim3(runs), dim3(64), 0, 0, g); }));
#define L16(D) [&] { hipLaunchKernelGGL((k_scan_pairl_asm<16, D>), dim3(runs), dim3(128), 48 * 1024, 0, l); }
#define L32(D) [&] { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_pairl_asm<32, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
                     hipLaunchKernelGGL((k_scan_pairl_asm<32, D>), dim3(runs), dim3(128), 96 * 1024, 0, l); }
    auto l16 = L16(0);
    auto l32 = L32(0);
    std::vector<int16_t> x1(halves), x2(halves);
    CK(hipMemcpy(x1.data(), dxs1, halves * 2, hipMemcpyDeviceToHost));
    auto check = [&](const char *name) {
        CK(hipMemcpy(x2.data(), dxs2, halves * 2, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < halves; ++i) bad += x1[i] != x2[i];
        printf("global-K vs %s outputs: %zu mismatches of %zu\n", name, bad, halves);
        CK(hipMemset(dxs2, 0, halves * 2));
    };
    report("pair, LDS-fed, 16 blocks per buffer", time_us(l16)); check("LDS-fed/16");
    report("pair, LDS-fed, 32 blocks per buffer", time_us(l32)); check("LDS-fed/32");
    report("  /16, helper without loads", time_us(L16(1))); report("  /32, helper without loads", time_us(L32(1)));
    report("  /16, helper only meets the barriers", time_us(L16(2))); report("  /32, helper only meets the barriers", time_us(L32(2)));
    return 0;
}
