// mfma_proj.hpp -- the projections of the fixed-point S5 layer on the int8 matrix cores (gfx950).
//
// w8a16: weights are <= 8 bit, activations <= 16 bit (SURVEY.md §8 / fxprun.py:302-308).  A 16-bit
// activation a is split into two signed byte planes
//       a = 256*hi + lo' + 128,     hi = a >> 8 (signed),  lo' = (a & 0xff) ^ 0x80 (signed byte)
// so   sum_k a_k w_k = 256*sum hi_k w_k + sum lo'_k w_k + 128*sum_k w_k
// is two passes of v_mfma_i32_32x32x32_i8 into one accumulator (shifted left by 8 in between) plus
// a per-column constant.  The accumulator is int32 and wraps (probed: tools/probe_mfma_i8.hip), and
// |sum| < 2^31 whenever K*2^15*2^7 < 2^31, so the result equals fxparray.py:662 bit for bit.
//
// This header holds what the MFMA kernels share: the packed-weight descriptor (MfmaW: [channel][k] int8 rows, row
// stride with an odd count of 16-byte slots, 128*colsum per channel), byte-plane and int16 pack/unpack helpers, the
// argument blocks of the encoder / decoder, and the element-wise kernels of the four-reduction BatchNorm path
// (k_bn_reduce16, k_resid16).  The kernels themselves are the phase-split ones in proj_p.hpp (encoder, B projection,
// decoder) and mfma_fused.hpp (C projection + gate): see proj_p.hpp for the design.
// Activations between kernels are int16 (N,H); the recurrence streams are int32 or int16 (scan_quad.hpp).
#pragma once
#include "s5fxp_kernels.hpp"

namespace s5 {

using v4i = __attribute__((ext_vector_type(4))) int;
using v2i = __attribute__((ext_vector_type(2))) int;
using v16i = __attribute__((ext_vector_type(16))) int;

struct MfmaW {
    const int8_t *wt;     // [Np][Kp] int8 (device), zero padded
    const int32_t *cs128; // [Np] 128 * sum_k w[k][ch]
    int32_t Kp, Np;       // row stride in bytes; padded channel count (multiple of 32)
};

__device__ __forceinline__ unsigned perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }

__device__ __forceinline__ v2i pack4_i16(int32_t a, int32_t b, int32_t c, int32_t d)
{
    v2i r;
    r[0] = (int)perm((unsigned)b, (unsigned)a, 0x05040100u);
    r[1] = (int)perm((unsigned)d, (unsigned)c, 0x05040100u);
    return r;
}

__device__ __forceinline__ void unpack4_i16(const v2i &w, int32_t (&v)[4])
{
    v[0] = (int32_t)(int16_t)(w[0] & 0xffff);
    v[1] = w[0] >> 16;
    v[2] = (int32_t)(int16_t)(w[1] & 0xffff);
    v[3] = w[1] >> 16;
}

// ---------------------------------------------------------------------------------------------
// Encoder: x int32 (N,K) -> relu(dense) int16 (N,H).  fxpmodel.py:331-366, 1263-1266.
// LDS: [weights Np*Kp][cs128 Np][bias_eff Np]
// ---------------------------------------------------------------------------------------------
struct EncArgs {
    const int32_t *x;
    int16_t *y;
    MfmaW w;
    const int32_t *bias_eff; // [Np] bias already moved to out_exp (fxparray.py:449-455)
    int64_t N;
    int32_t K, M;
    int32_t xb, xe, inp_bits, inp_exp, conv; // conversion of fxpmodel.py:335-347
    int32_t rs, out_bits;
    int32_t *status;
};

// ---------------------------------------------------------------------------------------------
// out2 dense + LUT sigmoid + mult_gate + maxima of the residual compute_best add.
// fxpmodel.py:1133-1137, 97-144, 1075-1093, 1147-1152.   LDS: [weights][cs128][bias_eff][lut 8]
// ---------------------------------------------------------------------------------------------
struct GateMArgs {
    const int16_t *x1;   // (N,H)
    const int16_t *skip; // (N,H) layer input
    int16_t *z;          // (N,H)
    MfmaW w;
    const int32_t *bias_eff;
    int32_t *tr_out2, *tr_sig, *tr_z; // optional int32 traces
    int64_t N;
    int32_t H;
    int32_t y_bits, y_exp, conv, inp_bits, inp_exp, rs, out_bits, out_exp;
    int32_t sig_x, sig_y;
    int32_t lut[8];
    int32_t l_bits, l_exp, r_bits, r_exp, res_bits, res_exp, rs_gate;
    DynExp skip_e;
    LayerDyn *dynw;
    const int32_t *run_if; // exact re-run: do the work only when *run_if != 0 (nullptr: always)
    int32_t mx_slot;       // first of the three LayerDyn::mx slots that receive the maxima
};

// ---------------------------------------------------------------------------------------------
// Decoder: int16 (N,H) with a device-chosen exponent -> int32 (N,M).  fxpmodel.py:1437, 331-366.
// CG column groups of NT tiles are processed one after the other from the same activation fragments.
// LDS: [weights][cs128][bias_eff]
// ---------------------------------------------------------------------------------------------
struct DecArgs {
    const int16_t *x;
    int32_t *y;
    MfmaW w;
    const int32_t *bias_eff;
    int64_t N;
    int32_t H, M;
    int32_t xb;
    DynExp xe;
    int32_t inp_bits, inp_exp, w_exp, out_bits, out_exp;
    int32_t *status;
};

// ---------------------------------------------------------------------------------------------
// element-wise pieces on int16 activations
// ---------------------------------------------------------------------------------------------
template <int STAGE>
__global__ __launch_bounds__(256) void k_bn_reduce16(BnArgs a, const int16_t *__restrict__ x, int64_t NH, int H, LayerDyn *dynw)
{
    const LayerDyn d = *a.dyn;
    const int xe = a.xe.get();
    float v[3] = {0.f, 0.f, 0.f};
    // 4 consecutive channels per thread (H % 4 == 0)
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < NH; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const int h0 = (int)(i % H);
        int32_t xv[4];
        unpack4_i16(*reinterpret_cast<const v2i *>(x + i), xv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int h = h0 + e;
            if (STAGE == 1) {
                const float fx = tofloat(xv[e], xe), fm = tofloat(a.mm[h], a.me);
                v[0] = fmaxf(v[0], fabsf(__fadd_rn(fx, fm)));
                v[1] = fmaxf(v[1], fabsf(fx));
                v[2] = fmaxf(v[2], fabsf(fm));
            } else if (STAGE == 2) {
                const int32_t t = bn_chain<1>(a, d, xv[e], h);
                v[0] = fmaxf(v[0], fabsf(__fmul_rn(tofloat(t, d.bn1.eo), tofloat(a.isv[h], a.ie))));
            } else if (STAGE == 3) {
                const int32_t t = bn_chain<2>(a, d, xv[e], h);
                v[0] = fmaxf(v[0], fabsf(__fmul_rn(tofloat(t, d.e2), tofloat(a.scale[h], a.se))));
            } else {
                const int32_t t = bn_chain<3>(a, d, xv[e], h);
                const float ft = tofloat(t, a.scale ? d.e3 : d.e2), fb = tofloat(a.bias[h], a.be);
                v[0] = fmaxf(v[0], fabsf(__fadd_rn(ft, fb)));
                v[1] = fmaxf(v[1], fabsf(ft));
                v[2] = fmaxf(v[2], fabsf(fb));
            }
        }
    }
    constexpr int slot = STAGE == 1 ? 0 : (STAGE == 2 ? 3 : (STAGE == 3 ? 4 : 5));
    if (STAGE == 1 || STAGE == 4) block_max_atomic<3>(v, dynw->mx + slot);
    else {
        float w[1] = {v[0]};
        block_max_atomic<1>(w, dynw->mx + slot);
    }
}

// residual add (compute_best) + ReLU on int16.  fxpmodel.py:1147-1159
__global__ __launch_bounds__(256) void k_resid16(const int16_t *__restrict__ z, const int16_t *__restrict__ skip,
                                                 int16_t *__restrict__ out, int32_t *tr_resid, int64_t NH, int res_bits,
                                                 int skip_bits, const LayerDyn *dyn)
{
    const AddCb p = dyn->res;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < NH; i += (int64_t)gridDim.x * blockDim.x * 4) {
        int32_t zv[4], sv[4], o[4];
        unpack4_i16(*reinterpret_cast<const v2i *>(z + i), zv);
        unpack4_i16(*reinterpret_cast<const v2i *>(skip + i), sv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int32_t rr = add_cb_apply(zv[e], res_bits, sv[e], skip_bits, p, res_bits);
            if (tr_resid) tr_resid[i + e] = rr;
            o[e] = rr < 0 ? 0 : rr;
        }
        *reinterpret_cast<v2i *>(out + i) = pack4_i16(o[0], o[1], o[2], o[3]);
    }
}

} // namespace s5
