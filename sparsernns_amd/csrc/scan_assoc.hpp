// scan_assoc.hpp -- the FLOAT model's diagonal-SSM scan (sparseRNNs/model/ssm.py:54-77 binary operator, :127
// jax.lax.associative_scan over (Lambda_elements, Bu_elements); reverse=True at :166-168 for the bidirectional half).
//
// x_t = A * x_{t-1} + Bu_t, complex64, A constant over time (ssm.py:106-108 tiles Lambda_bar over L).  The reference
// evaluates it as a parallel prefix over the pairs (A, Bu) with the operator (a_i, b_i) o (a_j, b_j) = (a_j a_i, a_j b_i + b_j).
// Floating-point prefix sums depend on the combination tree, so this path is TOLERANCE-checked (tests state the bound), not
// bit-exact; it is not the integer path (fxpmodel.py:430-432 asserts that one non-associative).
//
// Decomposition (time-parallel, one pass over HBM):
//   workgroup = one sequence x 16 states x all L steps; 8 waves; lane = (segment-in-wave s4 in 0..3) * 16 + state
//   chunk     = 32 segments (8 waves x 4) of SEG = 16 consecutive steps = 512 steps, held in registers (32 floats / lane)
//   per chunk: (1) fold each segment from zero: e = its aggregate under the operator (first component A^16, a constant);
//              (2) prefix over the 64 segment aggregates: two __shfl_up steps inside the wave (Hillis-Steele with the
//                  constants A^16, A^32), wave aggregates through LDS, a serial fold over the <= 7 earlier waves;
//              (3) run the recurrence again from the segment's carry-in over the same registers and store.
//   The carry between chunks is the fold over all 8 wave aggregates, computed redundantly by every lane.
// Loads / stores: a lane's step k is row t = t0 + 16 * segment + k, 16 states = 128 contiguous bytes; each wave
// instruction covers four such rows.  Bytes moved = 8 * P in + 8 * P out per frame and sequence: the algorithmic minimum.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace s5 {

struct cf32 {
    float re, im;
};
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf32 cfma(cf32 a, cf32 x, cf32 b) // a * x + b
{
    return {fmaf(a.re, x.re, fmaf(-a.im, x.im, b.re)), fmaf(a.re, x.im, fmaf(a.im, x.re, b.im))};
}
__device__ __forceinline__ cf32 cshfl_up(cf32 v, int d) { return {__shfl_up(v.re, d, 64), __shfl_up(v.im, d, 64)}; }

struct ScanAssocArgs {
    const float2 *lambda; // (P) complex64
    const float2 *bu;     // (B,L,P) complex64
    float2 *xs;           // (B,L,P)
    const float2 *x0;     // (B,P) state before the first step, or nullptr (zeros)
    float2 *x_last;       // (B,P) state after the last step, or nullptr
    int32_t B, L, P;
    int32_t reverse; // 1: scan from t = L-1 down to 0 (ssm.py:166-168)
};

constexpr int ASSOC_SEG = 16, ASSOC_WAVES = 8, ASSOC_PT = 16;
constexpr int ASSOC_CHUNK = ASSOC_SEG * 4 * ASSOC_WAVES; // 512 steps

__global__ __launch_bounds__(64 * ASSOC_WAVES) void k_scan_assoc_c64(ScanAssocArgs a)
{
    __shared__ cf32 agg[2][ASSOC_WAVES][ASSOC_PT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = lane & 15, s4 = lane >> 4;
    const int ptiles = (a.P + ASSOC_PT - 1) / ASSOC_PT;
    const int b = blockIdx.x / ptiles, p = (blockIdx.x % ptiles) * ASSOC_PT + pl;
    const bool live = p < a.P;
    const cf32 A = live ? cf32{a.lambda[p].x, a.lambda[p].y} : cf32{0.f, 0.f};
    // constants: A^16 (one segment), A^32, A^48 for the in-wave prefix, A^64 (one wave).  Squared in double and rounded once:
    // a power that is used for every carry acts like a perturbed pole, and its relative error is amplified by
    // 1 / (1 - |A|^64) in the states (complex64 squaring: 1.7e-5 * max|x| at |A| = 0.9995, L = 4096 -- the same as the
    // complex64 tree of the reference; this way 2e-6)
    double pr = A.re, pi = A.im;
    cf32 A16, A32, A48, A64;
#pragma unroll
    for (int q = 1; q <= 6; ++q) { // A^(2^q)
        const double r2 = pr * pr - pi * pi, i2 = 2.0 * pr * pi;
        pr = r2; pi = i2;
        if (q == 4) A16 = {(float)pr, (float)pi};
        if (q == 5) A32 = {(float)pr, (float)pi};
        if (q == 6) A64 = {(float)pr, (float)pi};
    }
    {
        const double r16 = (double)A16.re, i16 = (double)A16.im, r32 = (double)A32.re, i32 = (double)A32.im;
        A48 = {(float)(r16 * r32 - i16 * i32), (float)(r16 * i32 + i16 * r32)};
    }
    const cf32 Aseg = s4 == 0 ? cf32{1.f, 0.f} : s4 == 1 ? A16 : s4 == 2 ? A32 : A48; // A^(16 * s4)
    const ptrdiff_t step0 = (ptrdiff_t)a.P, step = a.reverse ? -step0 : step0; // elements per row; per scan step
    const size_t row = (size_t)a.P;
    const float2 *src = a.bu + (size_t)b * a.L * row + p;
    float2 *dst = a.xs + (size_t)b * a.L * row + p;
    cf32 carry = (live && a.x0) ? cf32{a.x0[(size_t)b * a.P + p].x, a.x0[(size_t)b * a.P + p].y} : cf32{0.f, 0.f};
    const int seg = wave * 4 + s4;
    int buf = 0;
    cf32 xlast{0.f, 0.f};
    bool have_last = false;
    for (int t0 = 0; t0 < a.L; t0 += ASSOC_CHUNK, buf ^= 1) {
        const int ts = t0 + seg * ASSOC_SEG; // first (scan-order) step of this lane's segment
        const ptrdiff_t off0 = (ptrdiff_t)(a.reverse ? a.L - 1 - ts : ts) * step0; // element offset of step ts
        cf32 u[ASSOC_SEG];
#pragma unroll
        for (int k = 0; k < ASSOC_SEG; ++k) {
            float2 v = make_float2(0.f, 0.f);
            if (live && ts + k < a.L) v = src[off0 + k * step];
            u[k] = {v.x, v.y};
        }
        // (1) segment aggregate
        cf32 e = u[0];
#pragma unroll
        for (int k = 1; k < ASSOC_SEG; ++k) e = cfma(A, e, u[k]);
        // (2) inclusive prefix over the four segments of this wave
        cf32 inc = e, o = cshfl_up(inc, 16);
        if (s4 >= 1) inc = cfma(A16, o, inc);
        o = cshfl_up(inc, 32);
        if (s4 >= 2) inc = cfma(A32, o, inc);
        if (s4 == 3) agg[buf][wave][pl] = inc; // the wave's aggregate (its first component is A^64)
        cf32 exc = cshfl_up(inc, 16);          // exclusive within the wave
        if (s4 == 0) exc = {0.f, 0.f};
        __syncthreads();
        cf32 cw = carry, call = carry; // carry into this wave / out of the chunk
#pragma unroll
        for (int w = 0; w < ASSOC_WAVES; ++w) {
            const cf32 g = agg[buf][w][pl];
            call = cfma(A64, call, g);
            if (w < wave) cw = call;
        }
        carry = call;
        // carry into this lane's segment: A^(16 s4) * cw + (aggregates of the earlier segments of this wave)
        cf32 x = cfma(Aseg, cw, exc);
        // (3) the recurrence from that carry
#pragma unroll
        for (int k = 0; k < ASSOC_SEG; ++k) {
            x = cfma(A, x, u[k]);
            if (live && ts + k < a.L) dst[off0 + k * step] = make_float2(x.re, x.im);
            if (ts + k == a.L - 1) { xlast = x; have_last = true; } // the lane that owns the last real step
        }
    }
    if (a.x_last && live && have_last) a.x_last[(size_t)b * a.P + p] = make_float2(xlast.re, xlast.im);
}

} // namespace s5
