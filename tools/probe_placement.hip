// probe_placement.hip -- on which SIMD does wave i of a six-wave workgroup run? (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_placement.hip -o tools/bin/probe_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(unsigned *out)
{
    extern __shared__ int lds[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127"); // 128 registers, as the tile kernels
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 50000) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6)] = hw;
    if (lds[threadIdx.x] == -1) out[0] = 1;
}

template <int THREADS>
void run(int wgs)
{
    const int W = THREADS / 64;
    unsigned *out; CK(hipMalloc(&out, (size_t)wgs * W * 4));
    hipLaunchKernelGGL(k<THREADS>, dim3(wgs), dim3(THREADS), 40 * 1024, 0, out);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h((size_t)wgs * W);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (+ XCC in another register)
    std::map<std::vector<int>, int> pat;
    for (int b = 0; b < wgs; ++b) {
        std::vector<int> p;
        for (int w = 0; w < W; ++w) p.push_back((h[(size_t)b * W + w] >> 4) & 3);
        pat[p]++;
    }
    printf("%d waves per workgroup, %d workgroups: SIMD of wave 0..%d -> count\n", W, wgs, W - 1);
    for (auto &kv : pat) { printf("   "); for (int s : kv.first) printf("%d ", s); printf(" x %d\n", kv.second); }
}

int main()
{
    run<384>(256);
    run<384>(512);
    run<256>(1024);
    run<512>(512);
    return 0;
}
