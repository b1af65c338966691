// probe_pk16.hip -- do the packed 16-bit instructions saturate the EXACT result? (diagnostic; decides what the gate kernel's
// epilogues may use).  hipcc --offload-arch=gfx950 -O3 tools/probe_pk16.hip -o tools/bin/probe_pk16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// every a in int16 (two per lane: a and ~a-ish partner), against a list of second / third operands
__global__ void k_probe(const int *bs, int nb, const int *cs, int nc, unsigned *bad)
{
    const int a0 = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 32768; // int16 range
    const int a1 = -a0 - 1;
    const unsigned ap = (unsigned)(a0 & 0xffff) | ((unsigned)(a1 & 0xffff) << 16);
    for (int ib = 0; ib < nb; ++ib)
        for (int ic = 0; ic < nc; ++ic) {
            const int b = bs[ib], c = cs[ic];
            const unsigned bp = (unsigned)(b & 0xffff) * 0x10001u, cp = (unsigned)(c & 0xffff) * 0x10001u;
            unsigned r;
            // 1. v_pk_mad_i16 clamp: sat16(a*b + c) on the exact value?
            asm volatile("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(ap), "v"(bp), "v"(cp));
            if ((short)(r & 0xffff) != sat16(a0 * b + c) || (short)(r >> 16) != sat16(a1 * b + c)) atomicAdd(bad + 0, 1);
            // 2. v_pk_sub_i16 clamp, v_pk_add_i16 clamp
            asm volatile("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(ap), "v"(cp));
            if ((short)(r & 0xffff) != sat16(a0 - c) || (short)(r >> 16) != sat16(a1 - c)) atomicAdd(bad + 1, 1);
            asm volatile("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(ap), "v"(cp));
            if ((short)(r & 0xffff) != sat16(a0 + c) || (short)(r >> 16) != sat16(a1 + c)) atomicAdd(bad + 2, 1);
            // 3. v_mul_i32_i24 with an SDWA half-word operand (sign-extended)
            int m0, m1;
            asm volatile("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(m0) : "v"(b), "v"(ap));
            asm volatile("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(m1) : "v"(b), "v"(ap));
            if (m0 != a0 * b || m1 != a1 * b) atomicAdd(bad + 3, 1);
            // 4. v_cvt_pk_i16_i32: saturating pack
            asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(r) : "v"(a0 * b + c), "v"(a1 * b + c));
            if ((short)(r & 0xffff) != sat16(a0 * b + c) || (short)(r >> 16) != sat16(a1 * b + c)) atomicAdd(bad + 4, 1);
            // 5. v_cvt_f32_i32 / v_ashrrev_i32 from an SDWA half
            float f0;
            int s1;
            asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f0) : "v"(ap));
            asm volatile("v_ashrrev_i32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(s1) : "v"(ib & 7), "v"(ap));
            if (f0 != (float)a1 || s1 != (a0 >> (ib & 7))) atomicAdd(bad + 5, 1);
            // 6. packed shifts and max
            asm volatile("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"((unsigned)(ib & 7) * 0x10001u), "v"(ap));
            if ((short)(r & 0xffff) != (short)(a0 >> (ib & 7)) || (short)(r >> 16) != (short)(a1 >> (ib & 7))) atomicAdd(bad + 6, 1);
            asm volatile("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(ap), "v"(cp));
            if ((short)(r & 0xffff) != (short)(a0 > (short)c ? a0 : (short)c) || (short)(r >> 16) != (short)(a1 > (short)c ? a1 : (short)c)) atomicAdd(bad + 7, 1);
        }
}

int main()
{
    std::vector<int> bs = {2, 1, 4, 8, 16, 256, -1, -2, 3, 32767, -32768, 127}, cs = {0, 1, -1, 5, -7, 32767, -32768, 12345, -12345, 16384, -16384, 255};
    int *db, *dc; unsigned *dbad;
    CK(hipMalloc(&db, bs.size() * 4)); CK(hipMalloc(&dc, cs.size() * 4)); CK(hipMalloc(&dbad, 64));
    CK(hipMemcpy(db, bs.data(), bs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dbad, 0, 64));
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, 0, db, (int)bs.size(), dc, (int)cs.size(), dbad);
    unsigned bad[8];
    CK(hipMemcpy(bad, dbad, 32, hipMemcpyDeviceToHost));
    const char *names[8] = {"v_pk_mad_i16 clamp == sat16(a*b+c)", "v_pk_sub_i16 clamp", "v_pk_add_i16 clamp", "v_mul_i32_i24_sdwa sext(WORD_n)",
                            "v_cvt_pk_i16_i32 saturates", "v_cvt_f32_i32_sdwa / v_ashrrev_i32_sdwa", "v_pk_ashrrev_i16", "v_pk_max_i16"};
    for (int i = 0; i < 8; ++i) printf("%-44s mismatches: %u\n", names[i], bad[i]);
    return 0;
}
