// ubench_mfma_i8.hip -- issue rate of the int8 MFMAs the projection kernels use (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_i8.hip -o tools/bin/ubench_mfma_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// MODE 0: one accumulation chain of 18 v_mfma_i32_32x32x32_i8 per iteration (the kernels' shape); 1: two chains of 9;
// 2: 16x16x64 (4 tiles x 18); BAR: a workgroup barrier per iteration
template <int MODE, bool BAR>
__global__ __launch_bounds__(384) void k(const int *in, int *out, unsigned long long *cyc, int iters)
{
    v4i a = *reinterpret_cast<const v4i *>(in + 4 * threadIdx.x), b = *reinterpret_cast<const v4i *>(in + 4 * threadIdx.x + 2048);
    v16i c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0;
    typedef int v4acc __attribute__((ext_vector_type(4)));
    v4acc d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 18; ++k) c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, c1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, a, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, a, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b, b, d3, 0, 0, 0);
            }
        }
        if (BAR) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    s += d0[0] + d1[1] + d2[2] + d3[3];
    if (s == 0x12345678) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char *name, K kern, int wgs, int threads, int mfma_per_iter)
{
    int *in, *out; unsigned long long *cyc;
    CK(hipMalloc(&in, 65536)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, wgs * 8));
    CK(hipMemset(in, 1, 65536));
    const int iters = 200;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, 0, in, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double waves_per_simd = (double)wgs * threads / 64 / 1024;
    printf("%-58s %4d WGs x %3d thr (%.2f waves/SIMD): %8.1f cycles/iteration/wave, %6.1f cycles per MFMA and SIMD, %7.1f us\n", name, wgs,
           threads, waves_per_simd, (double)c / iters, (double)c / iters / (mfma_per_iter * (waves_per_simd < 1 ? 1 : waves_per_simd)), ms * 1e3);
}

int main()
{
    run("32x32x32 i8, one chain of 18, 1 wave/SIMD", k<0, false>, 256, 256, 18);
    run("32x32x32 i8, one chain of 18, 3 waves/SIMD (2 WGs x 6 waves)", k<0, false>, 512, 384, 18);
    run("32x32x32 i8, one chain of 18, 3 waves/SIMD, barrier", k<0, true>, 512, 384, 18);
    run("32x32x32 i8, two chains of 9, 3 waves/SIMD, barrier", k<1, true>, 512, 384, 18);
    run("16x16x64 i8, four chains of 18, 3 waves/SIMD, barrier", k<2, true>, 512, 384, 72);
    return 0;
}
