// ubench_glds.hip -- how fast do rows of 257 int32 (1028 bytes: 4-byte-aligned row starts, the encoder's input) stream
// into a CU?  (diagnostic, not shipped)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_glds.hip -o tools/bin/ubench_glds
// Every workgroup (3 waves) walks 32-row tiles of a (N, 257) int32 matrix with a two-slot ring, the way an LDS-DMA
// encoder would: wait for slot s, barrier, issue the DMAs of the next tile into the other slot, "consume" slot s (read
// every row back from LDS and fold it into a checksum).  Variants: the DMA width (16 / 4 bytes), an aligned row
// stride (1024 + 16 bytes of padding in global memory) and plain global_load_dwordx4 into registers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define GLDS_SRC(p) ((const __attribute__((address_space(1))) void *)(p))
#define GLDS_DST(p) ((__attribute__((address_space(3))) void *)(p))

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int FT = 32, ROWB = 1040, NW = 3;

// MODE 0: glds x4 per row (+ one glds x1 for the tail column); 1: glds x1, five per row; 2: registers (global_load_dwordx4)
template <int MODE, int SLOTS>
__global__ __launch_bounds__(64 * NW) void k_stream(const int *x, long N, int stride, int *out, int consume)
{
    extern __shared__ __attribute__((aligned(16))) char raw[];
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long tiles = N / FT;
    int acc = 0;
    auto issue = [&](long tile, int slot) {
        for (int j = 0; j < 11; ++j) { // every wave issues the same number of DMAs (the last wave repeats row 31): counted waits
            int i = w + NW * j;
            i = i < FT ? i : FT - 1;
            const int *row = x + (tile * FT + i) * stride;
            char *dst = raw + (slot * FT + i) * ROWB;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds(GLDS_SRC(row + 4 * l), GLDS_DST(dst), 16, 0, 0);
            } else if (MODE == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) __builtin_amdgcn_global_load_lds(GLDS_SRC(row + 64 * c + l), GLDS_DST(dst + 256 * c), 4, 0, 0);
            }
        }
        if (MODE != 2) // the tail column of the 32 rows (every wave: equal counts)
            __builtin_amdgcn_global_load_lds(GLDS_SRC(x + (tile * FT + (l & 31)) * stride + 256), GLDS_DST(raw + SLOTS * FT * ROWB + slot * 256), 4, 0, 0);
    };
    long tile = blockIdx.x;
    if (MODE == 2) {
        // register staging: every wave loads its rows of the next tile before it consumes (stores to LDS) the current ones
        v4i r[11];
        auto fetch = [&](long t) {
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                int i = w + NW * j;
                i = i < FT ? i : FT - 1;
                r[j] = *reinterpret_cast<const v4i *>(x + (t * FT + i) * stride + 4 * l);
            }
        };
        if (tile < tiles) fetch(tile);
        for (; tile < tiles; tile += gridDim.x) {
#pragma unroll
            for (int j = 0; j < 11; ++j) {
                const int i = w + NW * j;
                if (i < FT) *reinterpret_cast<v4i *>(raw + i * ROWB + 16 * l) = r[j];
            }
            if (tile + gridDim.x < tiles) fetch(tile + gridDim.x);
            __syncthreads();
            if (consume)
                for (int i = w; i < FT; i += NW) { const v4i t = *reinterpret_cast<const v4i *>(raw + i * ROWB + 16 * l); acc += t[0] ^ t[1] ^ t[2] ^ t[3]; }
            __syncthreads();
        }
    } else {
        int slot = 0;
        for (int s = 0; s < SLOTS - 1; ++s)
            if (tile + s * gridDim.x < tiles) issue(tile + s * gridDim.x, s);
        for (; tile < tiles; tile += gridDim.x, slot = (slot + 1) % SLOTS) {
            // the oldest tile in flight has landed when at most (SLOTS - 2) tiles' worth of this wave's DMAs are outstanding
            constexpr int PER = (MODE == 0 ? 11 : 44) + 1; // DMAs per wave and tile
            if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (SLOTS == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PER) : "memory");
            const long nxt = tile + (long)(SLOTS - 1) * gridDim.x;
            if (nxt < tiles) issue(nxt, (slot + SLOTS - 1) % SLOTS);
            if (consume)
                for (int i = w; i < FT; i += NW) { const v4i t = *reinterpret_cast<const v4i *>(raw + (slot * FT + i) * ROWB + 16 * l); acc += t[0] ^ t[1] ^ t[2] ^ t[3]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
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

int main(int argc, char **argv)
{
    const long N = argc > 1 ? atol(argv[1]) : 131072;
    int *x, *out;
    CK(hipMalloc(&x, (size_t)N * 260 * 4 + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 1, (size_t)N * 260 * 4 + 4096));
    auto run = [&](const char *name, auto kernel, int stride, int slots, int wgs_per_cu, int consume) {
        const size_t smem = (size_t)slots * FT * ROWB + slots * 256;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const double us = time_us([&] { hipLaunchKernelGGL(kernel, dim3(256 * wgs_per_cu), dim3(64 * NW), smem, 0, x, N, stride, out, consume); });
        printf("%-64s stride %4d  %d slots x %d WG/CU  %7.1f us  %6.2f TB/s\n", name, stride * 4, slots, wgs_per_cu, us, (double)N * 1028 / us / 1e6);
    };
    for (int consume = 0; consume < 2; ++consume) {
        printf("--- %s\n", consume ? "rows read back from LDS" : "DMA only");
        run("LDS-DMA 16 B/lane, rows at 4-byte alignment", k_stream<0, 2>, 257, 2, 2, consume);
        run("LDS-DMA 16 B/lane, rows 16-byte aligned (padded matrix)", k_stream<0, 2>, 260, 2, 2, consume);
        run("LDS-DMA 4 B/lane, rows at 4-byte alignment", k_stream<1, 2>, 257, 2, 2, consume);
        run("LDS-DMA 16 B/lane, 4-byte alignment, 4 slots x 1 WG/CU", k_stream<0, 4>, 257, 4, 1, consume);
        run("LDS-DMA 16 B/lane, 4-byte alignment, 3 slots x 1 WG/CU", k_stream<0, 3>, 257, 3, 1, consume);
        run("LDS-DMA 16 B/lane, 4-byte alignment, 2 slots x 1 WG/CU", k_stream<0, 2>, 257, 2, 1, consume);
        run("registers (global_load_dwordx4 + ds_write), 4-byte alignment", k_stream<2, 1>, 257, 1, 2, consume);
        run("registers, 4 WG/CU", k_stream<2, 1>, 257, 1, 4, consume);
        run("registers, 16-byte aligned rows, 4 WG/CU", k_stream<2, 1>, 260, 1, 4, consume);
    }
    return 0;
}
