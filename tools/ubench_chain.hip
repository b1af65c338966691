// ubench_chain.hip -- instruction-issue experiments for the recurrence chain (diagnostic, not shipped).
//   python tools/gen_ubench_chain.py
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_chain.hip -o tools/bin/ubench_chain
// One wave per workgroup, 128 workgroups (the recurrence kernel's launch shape at B=32, P=64), 256 iterations of a
// 16-step body = 4096 steps; s_memtime around the loop.  Variants: tools/gen_ubench_chain.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ubench_chain_gen.inc"

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);       \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

struct Args {
    const int *in;   // per-wave input region, 1 MB each
    int *out;        // per-wave output region, 512 KB each
    const int *cst;  // 8 x 64 per-lane constants
    unsigned long long *cyc;
    int *sink;
    unsigned iters;
};

#define UB_KERNEL(NAME, NINS)                                                                                          \
    __global__ __launch_bounds__(64) void k_##NAME(Args a)                                                             \
    {                                                                                                                  \
        __shared__ int lds_buf[4096];                                                                                  \
        const int lane = threadIdx.x;                                                                                  \
        lds_buf[lane] = lane;                                                                                          \
        const unsigned wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);                                         \
        const unsigned long long bin = (unsigned long long)(a.in + (size_t)wave * (1u << 18)),                         \
                                 bout = (unsigned long long)(a.out + (size_t)wave * (1u << 17));                       \
        const unsigned r0 = (unsigned)bin, r1 = (unsigned)(bin >> 32) & 0xffffu, r2 = 1u << 20, r3 = 0x00020000u;      \
        const unsigned w0 = (unsigned)bout, w1 = (unsigned)(bout >> 32) & 0xffffu, w2 = 1u << 19;                      \
        int c2 = a.cst[lane], c3 = a.cst[64 + lane], c4 = a.cst[128 + lane], c5 = a.cst[192 + lane];                   \
        int x = a.cst[256 + lane];                                                                                     \
        const unsigned vo16 = lane * 16, vo8 = lane * 8;                                                               \
        unsigned long long t0, t1;                                                                                     \
        unsigned cnt = a.iters;                                                                                        \
        asm volatile("s_mov_b32 s8, %[r0]\n\ts_mov_b32 s9, %[r1]\n\ts_mov_b32 s10, %[r2]\n\ts_mov_b32 s11, %[r3]\n\t"   \
                     "s_mov_b32 s12, %[w0]\n\ts_mov_b32 s13, %[w1]\n\ts_mov_b32 s14, %[w2]\n\ts_mov_b32 s15, %[r3]\n\t" \
                     "s_mov_b32 s16, 0\n\ts_mov_b32 s17, 0\n\ts_mov_b32 s18, 0x1000\n\ts_mov_b32 s19, 0x800\n\t"       \
                     "v_mov_b32 v2, %[c2]\n\tv_mov_b32 v3, %[c3]\n\tv_mov_b32 v4, %[c4]\n\tv_mov_b32 v5, %[c5]\n\t"     \
                     "v_mov_b32 v9, %[vo16]\n\tv_mov_b32 v8, %[vo8]\n\tv_mov_b32 v7, %[vo16]\n\tv_mov_b32 v10, %[x]\n\t"                         \
                     "v_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 1\n\tv_mov_b32 v15, 2\n\t" \
                     "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"                 \
                     "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"                 \
                     "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\t"                                                        \
                     "s_nop 4\n\t"                                                                                     \
                     "s_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                     \
                     "1:\n\t" UB_BODY_##NAME                                                                           \
                     "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n\t"                  \
                     "s_memtime %[t1]\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                            \
                     "s_nop 4\n\tv_mov_b32 %[x], v10\n\t"                                                              \
                     : [t0] "=&s"(t0), [t1] "=&s"(t1), [cnt] "+s"(cnt), [x] "+v"(x)                                    \
                     : [r0] "s"(r0), [r1] "s"(r1), [r2] "s"(r2), [r3] "s"(r3), [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), \
                       [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [vo16] "v"(vo16), [vo8] "v"(vo8)         \
                     : "v2", "v3", "v4", "v5", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", \
                       "v23", "v24", "v25", "v26", "v27", "v28", "v29", "s8", "s9", "s10", "s11", "s12", "s13", "s14",  \
                       "s15", "s16", "s17", "s18", "s19", "memory", "scc");                                            \
        if (lane == 0) a.cyc[wave] = t1 - t0;                                                                          \
        if (x == 0x12345678) a.sink[0] = x + lds_buf[lane + 64];                                                                          \
    }

UB_ALL(UB_KERNEL)

template <class K>
void run(const char *name, int nins, K kernel, Args a, int waves)
{
    const int steps = (int)a.iters * 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, a);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(waves), dim3(64), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(waves);
    CK(hipMemcpy(c.data(), a.cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mx = 0, mn = ~0ull;
    double sum = 0;
    for (auto v : c) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; sum += (double)v; }
    const double us = ms * 1e3 / reps;
    printf("%-20s %5.2f ins/step  %8.2f us/launch  %6.2f ns/step | cycles/step: min %6.2f avg %6.2f max %6.2f | cycles/ins %5.2f\n",
           name, nins / 16.0, us, us * 1e3 / steps, (double)mn / steps, sum / waves / steps, (double)mx / steps,
           sum / waves / steps / (nins / 16.0));
}

int main()
{
    const int waves = 128;
    Args a{};
    int *in, *out, *cst, *sink;
    CK(hipMalloc(&in, (size_t)waves << 20));
    CK(hipMalloc(&out, (size_t)waves << 19));
    CK(hipMalloc(&cst, 8 * 64 * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&a.cyc, 4096 * 8));
    CK(hipMemset(in, 1, (size_t)waves << 20));
    std::vector<int> h(8 * 64);
    srand(3);
    for (int i = 0; i < 64; ++i) {
        h[i] = 60000 - 31 * i;         // c_own
        h[64 + i] = 2000 + 17 * i;     // c_partner
        h[128 + i] = 65534;            // k
        h[192 + i] = (i % 7) - 3;      // b
        h[256 + i] = 100 + i;          // x0
    }
    CK(hipMemcpy(cst, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    a.in = in; a.out = out; a.cst = cst; a.sink = sink; a.iters = 256;
#define UB_RUN(NAME, NINS) run(#NAME, NINS, k_##NAME, a, waves);
    UB_ALL(UB_RUN)
    return 0;
}
