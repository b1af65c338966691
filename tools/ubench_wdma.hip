// ubench_wdma.hip -- LDS-DMA streaming with WAVE-PRIVATE slots (the k_enc_w skeleton, no compute): how fast do rows of 257 int32
// stream into a CU when every wave runs its own two-slot ring and nothing is shared?  (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_wdma.hip -o tools/bin/ubench_wdma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define GLDS_SRC(p) ((const __attribute__((address_space(1))) void *)(p))
#define GLDS_DST(p) ((__attribute__((address_space(3))) void *)(p))

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int ROWB = 1040;

// FB rows per block, NW waves per workgroup, SLOTS slots per wave, TAIL: also gather element 256 of every row, ASMF: saddr + voffset
// form in hand-written asm (one statement per block) instead of the builtin with per-lane 64-bit addresses
template <int FB, int NW, int SLOTS, bool TAIL, bool ASMF>
__global__ __launch_bounds__(64 * NW) void k_wdma(const int *x, long N, int *out, int consume)
{
    extern __shared__ __attribute__((aligned(16))) char raw[];
    constexpr int SLOTB = FB * ROWB + 256;
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char *slots = raw + w * SLOTS * SLOTB;
    const long nblk = N / FB, nwaves = (long)gridDim.x * NW;
    int acc = 0;
    unsigned vo[FB];
    for (int r = 0; r < FB; ++r) vo[r] = r * 1028 + 16 * l;
    const unsigned vt = (l < FB ? l : FB - 1) * 1028 + 1024;
    auto issue = [&](long blk, char *slot) {
        const int *base = x + blk * FB * 257;
        if (ASMF) {
            const unsigned long long gpv = (unsigned long long)base;
            const unsigned long long gp = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gpv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)gpv);
            unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)slot);
#pragma unroll
            for (int r = 0; r < FB; ++r) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(vo[r]), "s"(gp) : "memory");
                lds += ROWB;
            }
            if (TAIL) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds), "v"(vt), "s"(gp) : "memory");
        } else {
#pragma unroll
            for (int r = 0; r < FB; ++r) __builtin_amdgcn_global_load_lds(GLDS_SRC(base + r * 257 + 4 * l), GLDS_DST(slot + r * ROWB), 16, 0, 0);
            if (TAIL) __builtin_amdgcn_global_load_lds(GLDS_SRC(base + (l < FB ? l : FB - 1) * 257 + 256), GLDS_DST(slot + FB * ROWB), 4, 0, 0);
        }
    };
    constexpr int PER = FB + (TAIL ? 1 : 0);
    long blk = (long)blockIdx.x * NW + w;
    for (int s = 0; s < SLOTS - 1; ++s)
        if (blk + s * nwaves < nblk) issue(blk + s * nwaves, slots + s * SLOTB);
    int slot = 0;
    for (; blk < nblk; blk += nwaves, slot = (slot + 1) % SLOTS) {
        const long nxt = blk + (long)(SLOTS - 1) * nwaves;
        if (nxt < nblk) issue(nxt, slots + ((slot + SLOTS - 1) % SLOTS) * SLOTB);
        // the oldest block in flight has landed when at most (SLOTS - 1) blocks' worth of DMAs are outstanding (tail iterations: fewer were issued -> over-waits, fine here)
        if (nxt < nblk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SLOTS - 1) * PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (consume) {
            const char *row = slots + slot * SLOTB + (l & 15) * ROWB + 64 * (l >> 4);
#pragma unroll
            for (int k = 0; k < FB; ++k) { const v4i t = *reinterpret_cast<const v4i *>(row + 16 * k); acc += t[0] ^ t[1] ^ t[2] ^ t[3]; }
            if (consume > 1) { // fake compute: a dependent VALU chain of `consume` x 64 instructions
                for (int i = 0; i < consume; ++i) {
#pragma unroll
                    for (int j = 0; j < 64; ++j) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(acc));
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) out[0] = acc;
}

template <class F>
double time_us(F launch, int reps = 10)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

// the same stream kernel timed alone (events around each launch) right after another kernel has WRITTEN `dirty_mb` MB
// elsewhere: does the streaming kernel pay for its predecessor's write-back?
__global__ void k_fill(int4 *p, long n16) { for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += gridDim.x * 256L) p[i] = int4{1, 2, 3, (int)i}; }

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 131072;
    int *x, *out;
    if (argc > 2) {
        const long dirty_mb = atol(argv[2]);
        int4 *d;
        CK(hipMalloc(&x, (size_t)N * 257 * 4 + 8192)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&d, (size_t)(dirty_mb > 0 ? dirty_mb : 1) << 20));
        CK(hipMemset(x, 1, (size_t)N * 257 * 4 + 8192));
        auto kernel = k_wdma<16, 4, 2, true, true>;
        const size_t smem = (size_t)4 * 2 * (16 * ROWB + 256);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double tot = 0, totf = 0;
        for (int rep = 0; rep < 12; ++rep) {
            hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
            CK(hipEventRecord(f0));
            if (dirty_mb > 0) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d, (long)dirty_mb << 16);
            CK(hipEventRecord(f1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kernel, dim3(256), dim3(256), smem, 0, x, N, out, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms, msf; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&msf, f0, f1));
            if (rep >= 2) { tot += ms; totf += msf; }
        }
        printf("stream kernel after a %ld MB fill: %.1f us per launch (%.2f TB/s of its own bytes); the fill itself %.1f us\n", dirty_mb, tot / 10 * 1e3,
               (double)N * 1028 / (tot / 10 * 1e3) / 1e6, totf / 10 * 1e3);
        return 0;
    }
    CK(hipMalloc(&x, (size_t)N * 257 * 4 + 8192)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 1, (size_t)N * 257 * 4 + 8192));
    auto run = [&](const char *name, auto kernel, int fb, int nw, int slots, int wgs_per_cu, int consume) {
        const size_t smem = (size_t)nw * slots * (fb * ROWB + 256);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const double us = time_us([&] { hipLaunchKernelGGL(kernel, dim3(256 * wgs_per_cu), dim3(64 * nw), smem, 0, x, N, out, consume); });
        printf("%-44s FB %2d  %d waves x %d slots x %d WG/CU (%3zu KB LDS/WG) consume %2d  %7.1f us  %6.2f TB/s\n", name, fb, nw, slots, wgs_per_cu, smem / 1024, consume, us,
               (double)N * 1028 / us / 1e6);
    };
    for (int consume : {0, 1, 8, 24}) {
        run("asm saddr, tail", k_wdma<16, 4, 2, true, true>, 16, 4, 2, 1, consume);
        run("asm saddr, no tail", k_wdma<16, 4, 2, false, true>, 16, 4, 2, 1, consume);
        run("builtin, tail", k_wdma<16, 4, 2, true, false>, 16, 4, 2, 1, consume);
        run("builtin, no tail", k_wdma<16, 4, 2, false, false>, 16, 4, 2, 1, consume);
        run("builtin, no tail, 8 waves x 1 slot", k_wdma<16, 8, 1, false, false>, 16, 8, 1, 1, consume);
        run("builtin, no tail, FB 8, 8 waves x 2 slots", k_wdma<8, 8, 2, false, false>, 8, 8, 2, 1, consume);
        run("builtin, no tail, FB 8, 4 waves x 2, 2 WG/CU", k_wdma<8, 4, 2, false, false>, 8, 4, 2, 2, consume);
        run("builtin, no tail, FB 16, 2 waves x 2, 2 WG/CU", k_wdma<16, 2, 2, false, false>, 16, 2, 2, 2, consume);
        run("builtin, no tail, FB 16, 4 waves x 3 slots?", k_wdma<8, 4, 4, false, false>, 8, 4, 4, 1, consume);
        run("builtin, tail, FB 32, 4 waves x 1 slot", k_wdma<32, 4, 1, true, false>, 32, 4, 1, 1, consume);
    }
    return 0;
}
