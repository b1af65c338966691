// ubench_scan.hip -- where do the cycles of the recurrence kernel go?  (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I sparsernns_amd/csrc tools/ubench_scan.hip -o /tmp/ubench_scan
// Variants of the quad step are timed with hipEvents (wall) and s_memtime (shader cycles) on the
// bench shape B=32, P=64, L=4096.
#include "scan_quad.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace s5;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);       \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

// MODE 0: full kernel (loads + stores); 1: no stores; 2: no loads/stores (pure chain);
// 3: pure chain without the DPP wait-state nop (wrong results, timing only); 4: chain of 3 plain v_add (issue floor)
template <int DEPTH, int MODE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_var(ScanQuadArgs a, unsigned long long *cyc)
{
    const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (BLOCK / 64) + wave_in_block);
    const int groups = a.P >> 4;
    const int b = wave / groups, p0 = (wave % groups) << 4;
    if (b >= a.B) return;
    const int s = lane >> 2, r = lane & 3;
    const int p = p0 + s;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int sre = 16 - a.ea_re, sim = 16 - a.ea_im;
    const int32_t kre = (1 << 16) - (1 << sre);
    int32_t cA, cB, kA = 0, kB = 0;
    if (r == 0) { cA = cB = Ar << sre; }
    else if (r == 1) { cA = cB = Ar << sim; }
    else if (r == 2) { cA = -(Ai << sre); kA = kre; cB = Ai << sim; }
    else { cA = Ai << sim; cB = -(Ai << sre); kB = kre; }
    const size_t wave_off = (((size_t)b * a.TB) * a.P + p0) * 8;
    const unsigned blk_stride = (unsigned)a.P * 32u;
    const unsigned extent = (unsigned)a.TB * blk_stride;
    auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.bq) + wave_off, 0, extent, 0x00020000);
    auto rout = __builtin_amdgcn_make_buffer_rsrc(a.xs + wave_off, 0, extent, 0x00020000);
    const unsigned voff = r < 2 ? (unsigned)(s * 32 + r * 16) : 0xFFFFFF00u;
    u32x4 ring[DEPTH];
    unsigned soff_ld = 0, soff_st = 0;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
        ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, soff_ld, 0);
        soff_ld += blk_stride;
    }
    int32_t x = 0, tmp = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int tb0 = 0; tb0 < a.TB; tb0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const u32x4 cur = ring[i];
            int32_t x1, x2, x3, x4;
            if (MODE == 3) {
#define STEP_NONOP(PERM, XIN, XOUT, C, K, BQ)                                                                   \
    asm volatile("v_mad_i32_i24 %0, %2, %3, %4\n\t"                                                             \
                 "v_add_u32_sdwa %0, sext(%0), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 "         \
                 "src1_sel:DWORD\n\t"                                                                           \
                 "v_add_u32 %1, %0, %0\n\t"                                                                     \
                 : "=&v"(tmp), "=v"(XOUT)                                                                       \
                 : "v"(C), "v"(XIN), "v"(K), "v"(BQ))
                STEP_NONOP("", x, x1, cA, kA, cur[0]);
                STEP_NONOP("", x1, x2, cB, kB, cur[1]);
                STEP_NONOP("", x2, x3, cA, kA, cur[2]);
                STEP_NONOP("", x3, x4, cB, kB, cur[3]);
            } else if (MODE == 5 || MODE == 6 || MODE == 7) {
#define STEP_V(MID, LAST, XIN, XOUT, C, K, BQ)                                                                  \
    asm volatile("v_mad_i32_i24 %0, %2, %3, %4\n\t"                                                             \
                 "v_add_u32_sdwa %0, sext(%0), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 "         \
                 "src1_sel:DWORD\n\t" MID LAST                                                                  \
                 : "=&v"(tmp), "=v"(XOUT)                                                                       \
                 : "v"(C), "v"(XIN), "v"(K), "v"(BQ))
#define DPPADD "v_add_u32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
#define PLAINADD "v_add_u32 %1, %0, %0\n\t"
                if (MODE == 5) {
                    STEP_V("s_nop 1\n\t", PLAINADD, x, x1, cA, kA, cur[0]); STEP_V("s_nop 1\n\t", PLAINADD, x1, x2, cB, kB, cur[1]);
                    STEP_V("s_nop 1\n\t", PLAINADD, x2, x3, cA, kA, cur[2]); STEP_V("s_nop 1\n\t", PLAINADD, x3, x4, cB, kB, cur[3]);
                } else if (MODE == 6) {
                    STEP_V("", DPPADD, x, x1, cA, kA, cur[0]); STEP_V("", DPPADD, x1, x2, cB, kB, cur[1]);
                    STEP_V("", DPPADD, x2, x3, cA, kA, cur[2]); STEP_V("", DPPADD, x3, x4, cB, kB, cur[3]);
                } else {
                    STEP_V("s_nop 0\n\t", DPPADD, x, x1, cA, kA, cur[0]); STEP_V("s_nop 0\n\t", DPPADD, x1, x2, cB, kB, cur[1]);
                    STEP_V("s_nop 0\n\t", DPPADD, x2, x3, cA, kA, cur[2]); STEP_V("s_nop 0\n\t", DPPADD, x3, x4, cB, kB, cur[3]);
                }
            } else if (MODE == 8 || MODE == 9) {
                // lane-per-state shapes (timing only): 8 = 4 mad + 2 sdwa-add + 2 add; 9 = 4 mad + 2 sdwa-add
                int32_t xi_ = tmp;
#define LANE8(XR, XI, XRO, XIO, BR, BI)                                                                         \
    asm volatile("v_mad_i32_i24 %2, %6, %4, %8\n\t"                                                             \
                 "v_mad_i32_i24 %3, %7, %5, %9\n\t"                                                             \
                 "v_mad_i32_i24 %0, %6, %5, %8\n\t"                                                             \
                 "v_mad_i32_i24 %1, %7, %4, %9\n\t"                                                             \
                 "v_add_u32_sdwa %2, sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t" \
                 "v_add_u32_sdwa %3, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t" \
                 "v_add_u32 %0, %2, %10\n\t"                                                                    \
                 "v_add_u32 %1, %3, %11\n\t"                                                                    \
                 : "=&v"(XRO), "=&v"(XIO), "=&v"(t2), "=&v"(t3)                                                 \
                 : "v"(XR), "v"(XI), "v"(cA), "v"(cB), "v"(kA), "v"(kB), "v"(BR), "v"(BI))
#define LANE6(XR, XI, XRO, XIO, BR, BI)                                                                         \
    asm volatile("v_mad_i32_i24 %2, %6, %4, %10\n\t"                                                            \
                 "v_mad_i32_i24 %3, %7, %5, %9\n\t"                                                             \
                 "v_mad_i32_i24 %0, %6, %5, %11\n\t"                                                            \
                 "v_mad_i32_i24 %1, %7, %4, %9\n\t"                                                             \
                 "v_add_u32_sdwa %0, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t" \
                 "v_add_u32_sdwa %1, sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n\t" \
                 : "=&v"(XRO), "=&v"(XIO), "=&v"(t2), "=&v"(t3)                                                 \
                 : "v"(XR), "v"(XI), "v"(cA), "v"(cB), "v"(kA), "v"(kB), "v"(BR), "v"(BI))
                int32_t t2, t3, y1, y2, y3, y4;
                if (MODE == 8) {
                    LANE8(x, xi_, x1, y1, cur[0], cur[1]); LANE8(x1, y1, x2, y2, cur[1], cur[2]);
                    LANE8(x2, y2, x3, y3, cur[2], cur[3]); LANE8(x3, y3, x4, y4, cur[3], cur[0]);
                } else {
                    LANE6(x, xi_, x1, y1, cur[0], cur[1]); LANE6(x1, y1, x2, y2, cur[1], cur[2]);
                    LANE6(x2, y2, x3, y3, cur[2], cur[3]); LANE6(x3, y3, x4, y4, cur[3], cur[0]);
                }
                tmp = y4;
            } else if (MODE == 4) {
#define STEP_ADD(XIN, XOUT, C, K, BQ)                                                                           \
    asm volatile("v_add_u32 %0, %2, %3\n\t"                                                                     \
                 "v_add_u32 %0, %0, %5\n\t"                                                                     \
                 "v_add_u32 %1, %0, %4\n\t"                                                                     \
                 : "=&v"(tmp), "=v"(XOUT)                                                                       \
                 : "v"(C), "v"(XIN), "v"(K), "v"(BQ))
                STEP_ADD(x, x1, cA, kA, cur[0]);
                STEP_ADD(x1, x2, cB, kB, cur[1]);
                STEP_ADD(x2, x3, cA, kA, cur[2]);
                STEP_ADD(x3, x4, cB, kB, cur[3]);
            } else {
                S5_SCAN_STEP("s_nop 1\n\t", "[2,3,0,1]", x, x1, cA, kA, cur[0]);
                S5_SCAN_STEP("", "[3,2,1,0]", x1, x2, cB, kB, cur[1]);
                S5_SCAN_STEP("", "[2,3,0,1]", x2, x3, cA, kA, cur[2]);
                S5_SCAN_STEP("", "[3,2,1,0]", x3, x4, cB, kB, cur[3]);
            }
            x = x4;
            if (MODE < 2) {
                ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, soff_ld < extent ? soff_ld : extent - blk_stride, 0);
                soff_ld += blk_stride;
            }
            if (MODE == 0) {
                u32x4 o;
                o[0] = (unsigned)x1; o[1] = (unsigned)x2; o[2] = (unsigned)x3; o[3] = (unsigned)x4;
                __builtin_amdgcn_raw_buffer_store_b128(o, rout, voff, soff_st, 0);
                soff_st += blk_stride;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[wave] = t1 - t0;
    if (MODE != 0 && (x ^ tmp) == 0x12345678) a.xs[0] = x; // keep the chain alive
}

template <int DEPTH, int MODE, int BLOCK>
void run(const char *name, ScanQuadArgs a, unsigned long long *dcyc, int waves, int L)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = (waves + BLOCK / 64 - 1) / (BLOCK / 64);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_var<DEPTH, MODE, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, a, dcyc);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_var<DEPTH, MODE, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, a, dcyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(waves);
    CK(hipMemcpy(c.data(), dcyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0, mn = ~0ull;
    double sum = 0;
    for (auto v : c) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; sum += (double)v; }
    const double us = ms * 1e3 / reps;
    printf("%-44s %8.2f us/launch  %6.2f ns/step | s_memtime/step: min %.2f avg %.2f max %.2f  (ticks: 100 MHz => x10ns)\n", name, us,
           us * 1e3 / L, (double)mn / L, sum / waves / L, (double)mx / L);
}

int main()
{
    const int B = 32, P = 64, L = 4096, TB = L / 4;
    const size_t words = (size_t)B * TB * P * 8;
    int32_t *bq, *xs, *are, *aim;
    unsigned long long *dcyc;
    CK(hipMalloc(&bq, words * 4 + 64 * P * 32));
    CK(hipMalloc(&xs, words * 4));
    CK(hipMalloc(&are, P * 4));
    CK(hipMalloc(&aim, P * 4));
    CK(hipMalloc(&dcyc, 4096 * 8));
    std::vector<int32_t> h(words), ar(P), ai(P);
    srand(1);
    for (auto &v : h) v = (rand() % 2001) - 1000;
    for (int p = 0; p < P; ++p) { ar[p] = 30000 - 37 * p; ai[p] = (p % 2 ? 1 : -1) * (1000 + 91 * p); }
    CK(hipMemcpy(bq, h.data(), words * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(are, ar.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(aim, ai.data(), P * 4, hipMemcpyHostToDevice));
    ScanQuadArgs a{bq, xs, are, aim, B, TB, P, 15, 15};
    const int waves = B * P / 16;
    run<8, 0, 256>("full d8 block256", a, dcyc, waves, L);
    run<8, 0, 64>("full d8 block64", a, dcyc, waves, L);
    run<16, 0, 64>("full d16 block64", a, dcyc, waves, L);
    run<4, 0, 64>("full d4 block64", a, dcyc, waves, L);
    run<8, 1, 64>("no stores d8 block64", a, dcyc, waves, L);
    run<8, 2, 64>("pure chain (mad,sdwa,nop,dpp) block64", a, dcyc, waves, L);
    run<8, 3, 64>("pure chain no nop / no dpp block64", a, dcyc, waves, L);
    run<8, 4, 64>("3 dependent v_add block64", a, dcyc, waves, L);
    run<8, 2, 256>("pure chain block256", a, dcyc, waves, L);
    run<8, 5, 64>("mad,sdwa,nop1,plain add", a, dcyc, waves, L);
    run<8, 6, 64>("mad,sdwa,dpp add (no nop; timing only)", a, dcyc, waves, L);
    run<8, 7, 64>("mad,sdwa,nop0,dpp add (timing only)", a, dcyc, waves, L);
    run<8, 8, 64>("lane-per-state 8 op", a, dcyc, waves, L);
    run<8, 9, 64>("lane-per-state 6 op (restricted)", a, dcyc, waves, L);
    // hand-scheduled kernel: same results as the compiler-scheduled one?  how fast?
    std::vector<int32_t> r0(words), r1(words);
    hipLaunchKernelGGL((k_var<8, 0, 64>), dim3(waves), dim3(64), 0, 0, a, dcyc);
    CK(hipMemcpy(r0.data(), xs, words * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(xs, 0xff, words * 4));
    hipLaunchKernelGGL(k_scan_quad_asm, dim3(waves), dim3(64), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r1.data(), xs, words * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < words; ++i) bad += r0[i] != r1[i];
    printf("asm kernel vs compiler-scheduled kernel: %zu mismatches of %zu words\n", bad, words);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_scan_quad_asm, dim3(waves), dim3(64), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us/launch  %6.2f ns/step  => %.1f GB/s algorithmic (%.1f%% of 8 TB/s)\n", "hand-scheduled asm d16 block64", ms * 50.0,
           ms * 50.0 * 1e3 / L, (double)B * L * P * 16 / (ms * 50.0 * 1e-6) / 1e9, (double)B * L * P * 16 / (ms * 50.0 * 1e-6) / 8e12 * 100);
    return 0;
}
