// probe_occupancy.hip -- how many 6-wave workgroups fit on a CU at a given VGPR count? (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_occupancy.hip -o tools/bin/probe_occupancy
// Every workgroup spins for a fixed number of cycles; PER workgroups per CU are launched; if they all fit the launch
// takes one spin, otherwise two or more.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int VGPRS, int THREADS>
__global__ __launch_bounds__(THREADS) void k_spin(int *out, long long cycles)
{
    extern __shared__ int lds[];
    if (VGPRS == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (VGPRS == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (VGPRS == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == -1) out[0] = 1;
}

template <class K>
void run(const char *name, K kern, int threads, int lds_kb, int per)
{
    int *out; CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long cycles = 100000;
    hipLaunchKernelGGL(kern, dim3(256 * per), dim3(threads), lds_kb * 1024, 0, out, cycles);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256 * per), dim3(threads), lds_kb * 1024, 0, out, cycles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %3d threads, %3d KB LDS, %d workgroups per CU launched: %7.1f us (one spin = ~%.0f us)\n", name, threads, lds_kb, per, ms * 1e3,
           cycles / 2400.0);
}

int main()
{
    for (int per = 2; per <= 4; ++per) {
        run("6 waves,  80 VGPRs", k_spin<80, 384>, 384, 40, per);
        run("6 waves,  96 VGPRs", k_spin<96, 384>, 384, 40, per);
        run("6 waves, 104 VGPRs", k_spin<104, 384>, 384, 40, per);
        run("6 waves, 128 VGPRs", k_spin<128, 384>, 384, 40, per);
        run("4 waves, 128 VGPRs", k_spin<128, 256>, 256, 30, per);
        run("4 waves,  96 VGPRs", k_spin<96, 256>, 256, 30, per);
    }
    return 0;
}
