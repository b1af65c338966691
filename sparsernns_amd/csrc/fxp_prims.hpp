// fxp_prims.hpp -- device-side integer primitives of the fixed-point S5 path (gfx950).
//
// Each function states the reference lines it reproduces (paths are into
// /root/reference/sparseRNNs/).  Everything is int32 with two's-complement wrap, which is
// what the reference computes under default JAX (x64 disabled).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fxp {

__host__ __device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__host__ __device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__host__ __device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__host__ __device__ __forceinline__ int32_t wshl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
__host__ __device__ __forceinline__ int32_t asr(int32_t a, int s) { return a >> s; }

// fxp_clip, fxparray.py:329-334,346-357 (signed).  On the device ONE v_med3_i32: the compiler only forms
// med3 from min/max with constant bounds, so it is spelled out (bounds are uniform and hoisted).
__host__ __device__ __forceinline__ int32_t sat(int32_t v, int bits)
{
    const int32_t hi = (int32_t)((1u << (bits - 1)) - 1u);
    const int32_t lo = ~hi;
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#else
    return v > hi ? hi : (v < lo ? lo : v);
#endif
}

// The same clip with its two bounds resident in VGPRs.  sat(v, bits) hands v_med3_i32 two wave-uniform bounds; the
// instruction takes at most one scalar operand, and the compiler re-creates the vector copies from the scalar registers in
// front of EVERY use (two v_mov per clip: 399 of the 1 099 vector instructions of the B projection's loop body) rather than
// keep them live.  sat_bounds() makes the copies once, through a statement the compiler can neither repeat nor fold; kernels
// call it before their loops for the widths their inner loops clip to.
struct SatB {
    int32_t lo, hi;
};
__host__ __device__ __forceinline__ SatB sat_bounds(int bits) // bits wave-uniform
{
    const int32_t hi = (int32_t)((1u << (bits - 1)) - 1u), lo = ~hi;
    SatB b;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(b.lo), "=v"(b.hi) : "s"(lo), "s"(hi));
#else
    b.lo = lo; b.hi = hi;
#endif
    return b;
}
__host__ __device__ __forceinline__ int32_t sat(int32_t v, const SatB &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(b.lo), "v"(b.hi));
    return r;
#else
    return v > b.hi ? b.hi : (v < b.lo ? b.lo : v);
#endif
}

// fxp_change_exp, fxparray.py:310-326: unchanged exponent -> NO clip; otherwise shift and clip
// at the operand's CURRENT bits.  Branch-free: one of the two shifts is by zero, and "no clip" is a clip at
// 32 bits (the identity).
__host__ __device__ __forceinline__ int32_t chexp(int32_t d, int bits, int e, int e2)
{
    const int l = e2 > e ? e2 - e : 0, r = e > e2 ? e - e2 : 0;
    return sat(asr(wshl(d, l), r), e2 == e ? 32 : bits);
}

// fxp_change_cfg, fxparray.py:232-271 (signed, FLOOR): change_exp, then a clip at the new bits if they are
// fewer.  Two nested symmetric clips are one clip at the smaller width.
__host__ __device__ __forceinline__ int32_t chcfg(int32_t d, int bits, int e, int bits2, int e2)
{
    const int l = e2 > e ? e2 - e : 0, r = e > e2 ? e - e2 : 0;
    const int b1 = e2 == e ? 32 : bits, b2 = bits > bits2 ? bits2 : 32;
    return sat(asr(wshl(d, l), r), b1 < b2 ? b1 : b2);
}

// FxpArray.to_float, fxparray.py:72-73: int32 -> f32 (RNE) then an exact power-of-two scale.
__device__ __forceinline__ float tofloat(int32_t d, int e) { return ldexpf((float)d, -e); }

// f32 -> s32 as XLA converts: truncate toward zero, saturate (v_cvt_i32_f32 does exactly that).
__device__ __forceinline__ int32_t f2i(float f) { return (int32_t)f; }

// Complex ReLU, fxpmodel.py:30-45: jax.nn.relu on complex64 == lexicographic maximum(z, 0);
// both parts round-trip through float32.
__device__ __forceinline__ void crelu(int32_t &re, int32_t &im)
{
    const float fr = (float)re, fi = (float)im;
    const bool keep = (fr > 0.f) || (fr == 0.f && fi > 0.f);
    re = keep ? f2i(fr) : 0;
    im = keep ? f2i(fi) : 0;
}

// FxpSigmoid.apply + lut_sigmoid_half, fxpmodel.py:97-144.  lut has 8 entries.
__device__ __forceinline__ int32_t sigmoid_lut(int32_t x, int xbits, int xe, int sx, int sy, const int32_t *lut)
{
    const int32_t xx = chexp(x, xbits, xe, sx);
    const int32_t a = xx < 0 ? wsub(0, xx) : xx;
    int32_t ind = asr(a, sx);
    ind = ind > 6 ? 6 : ind;
    const int32_t mu = a & ((1 << sx) - 1);
    const int32_t half = wadd(asr(wmul((1 << sx) - mu, lut[ind]), sx), asr(wmul(mu, lut[ind + 1]), sx));
    return wadd(1 << (sy - 1), xx > 0 ? half : wsub(0, half));
}

// max(0, int(ceil(log2(m + eps)))) in float32 with a correctly rounded log2 (fxparray.py:421-425,
// 603-607).  The hardware log2 (v_log_f32, about 1 ulp) decides whenever the result is not within 1e-3 of an
// integer -- there neither its error nor the rounding of the true log2 to float32 can move the ceiling; only
// next to an integer the (slow, software) double log2 is evaluated and rounded.  The finalize code that calls
// this sits at the head and tail of the residual pass, where a few microseconds per workgroup are visible.
__device__ inline int intbits_f32(float m, float eps)
{
    const float v = __fadd_rn(m, eps);
    float l = __log2f(v);
    const float fr = l - floorf(l);
    if (!(fr > 1e-3f && fr < 1.f - 1e-3f)) l = (float)log2((double)v);
    const int c = (int)ceilf(l);
    return c > 0 ? c : 0;
}

// ---- "compute_best" add in device-exponent form (fxparray.py:420-448) ------------------------
struct AddCb {
    int32_t shx;  // left shift that brings x to agg_exp (>= 0)
    int32_t shy;  // same for y
    int32_t post; // result_exp - agg_exp : > 0 left shift, < 0 arithmetic right shift
    int32_t eo;   // result_exp
};

__device__ __forceinline__ int32_t add_cb_apply(int32_t x, int xb, int32_t y, int yb, const AddCb &p, int ob)
{
    // change_cfg(op -> max(agg_bits, bits), agg_exp): a left shift saturates at the operand's own
    // bits (fxparray.py:321-325); the widening afterwards never clips.
    const int32_t a = p.shx ? sat(wshl(x, p.shx), xb) : x;
    const int32_t b = p.shy ? sat(wshl(y, p.shy), yb) : y;
    int32_t s = wadd(a, b);
    s = p.post > 0 ? wshl(s, p.post) : (p.post < 0 ? asr(s, -p.post) : s);
    return sat(s, ob);
}

// add_cb_apply with the operands' and the result's clip bounds resident in VGPRs (sat_bounds) and the uniform shift decisions
// taken once: what the residual pass and the decoder's fused residual run per element
struct AddCbV {
    int32_t shx, shy, lsh, rsh; // left shifts of the operands to agg_exp; the result's left / right shift
    SatB sx, sy, so;
};
__device__ __forceinline__ AddCbV make_add_cb_v(const AddCb &p, int xb, int yb, int ob)
{
    AddCbV v;
    v.shx = p.shx; v.shy = p.shy; v.lsh = p.post > 0 ? p.post : 0; v.rsh = p.post < 0 ? -p.post : 0;
    v.sx = sat_bounds(p.shx ? xb : 32); v.sy = sat_bounds(p.shy ? yb : 32); v.so = sat_bounds(ob);
    return v;
}
__device__ __forceinline__ int32_t add_cb_apply(int32_t x, int32_t y, const AddCbV &p)
{
    // a shift of 0 with 32-bit bounds is the identity: the same value as the branches of add_cb_apply above
    const int32_t a = sat(wshl(x, p.shx), p.sx), b = sat(wshl(y, p.shy), p.sy);
    return sat(asr(wshl(wadd(a, b), p.lsh), p.rsh), p.so);
}

} // namespace fxp
