// probe_mfma_i8.hip -- checks the operand lane maps of v_mfma_i32_32x32x32_i8 and unaligned 16-byte global access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;
// D[32 x 32] = A[32 x 32(k)] * B[32(k) x 32]; assumed maps: A lane l holds A[l&31][16*(l>>5) + j], B lane l holds B[16*(l>>5)+j][l&31]
__global__ void k(const int8_t* A /*[32][32] row-major*/, const int8_t* Bt /*[col][k]*/, int* D /*[32][32]*/, int big)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a = *reinterpret_cast<const v4i*>(A + r * 32 + 16 * h);
    v4i b = *reinterpret_cast<const v4i*>(Bt + r * 32 + 16 * h);
    v16i c;
    for (int i = 0; i < 16; ++i) c[i] = big;
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    if (big) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r;
        D[row * 32 + col] = c[i];
    }
}
__global__ void k_unaligned(const int* src, int* dst, int n)
{
    // rows of 257 ints: 16-byte accesses at 4-byte alignment
    const int r = blockIdx.x, t = threadIdx.x;
    if (t * 4 + 4 <= 257) {
        v4i v = *reinterpret_cast<const v4i*>(src + r * 257 + t * 4);
        *reinterpret_cast<v4i*>(dst + r * 257 + t * 4) = v;
    }
}
int main()
{
    std::vector<int8_t> A(1024), Bt(1024);
    srand(3);
    for (auto& v : A) v = (int8_t)(rand() % 256 - 128);
    for (auto& v : Bt) v = (int8_t)(rand() % 256 - 128);
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 1024, hipMemcpyHostToDevice);
    for (int big : {0, 0x7ffffff0}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, big);
        std::vector<int> D(1024);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            long long s = 0;
            for (int kk = 0; kk < 32; ++kk) s += (long long)A[i * 32 + kk] * Bt[j * 32 + kk];
            long long exp = big ? (long long)big + 2 * s : s;
            int want = (int)(unsigned)(unsigned long long)exp; // wrap
            if (D[i * 32 + j] != want) { if (bad < 5) printf("mismatch [%d][%d]: got %d want(wrap) %d (true %lld)\n", i, j, D[i*32+j], want, exp); ++bad; }
        }
        printf("mfma_i32_32x32x32_i8 with C=%d: %d mismatches vs wrap-around reference\n", big, bad);
    }
    const int R = 64; std::vector<int> src(R * 257 + 8), dst(R * 257 + 8, -1);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (int)i * 7 + 1;
    int *ds, *dd; hipMalloc(&ds, src.size() * 4); hipMalloc(&dd, dst.size() * 4);
    hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice); hipMemset(dd, 0xff, dst.size() * 4);
    hipLaunchKernelGGL(k_unaligned, dim3(R), dim3(64), 0, 0, ds, dd, 0);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(dst.data(), dd, dst.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int r = 0; r < R; ++r) for (int c = 0; c < 256; ++c) bad += dst[r * 257 + c] != src[r * 257 + c];
    printf("unaligned 16-byte load/store (4-byte aligned rows of 257 ints): %s, %d mismatches\n", hipGetErrorString(e), bad);
    return 0;
}
