// mfma_bn.hpp -- BatchNorm exponents from per-channel extremes, and the lean B projection.
//
// (1) The four BatchNorm ops pick their exponents from float32 maxima over the whole (B,L,H) tensor
//     (fxpmodel.py:892-933, fxparray.py:421-432,602-608).  Per channel h every stage is a MONOTONE map of
//     the layer input x (shift, saturate, add a constant, multiply by a constant, floor shift -- no int32
//     wrap while all operands are <= 16 bit with exponents in [0,15], which the host checks), int32->float32
//     and float32 add/mul by a constant are monotone as well, and |.| of a monotone function peaks at an end
//     point.  So max over the tensor = max over channels of the two end points min_h(x), max_h(x): ONE pass
//     over the tensor (k_resid_minmax16, fused with the previous layer's residual add) and one single-workgroup kernel (k_bn_finalize_mm) replace four reduction
//     passes.  The results are identical by construction; tests compare against the oracle's full reductions.
//
// (2) the lean 16-bit BatchNorm chain (Bn16) and the argument block of the B projection (kernel: proj_p.hpp
//     k_bproj_p): BatchNorm parameters come from LDS, the chain runs four elements at a time, and the SSM input u
//     is written once (int16) so the C projection does not recompute the chain.
#pragma once
#include "mfma_proj.hpp"

namespace s5 {

// one launch instead of two memsets at the head of a forward: the status words and the per-layer device state.  What the
// host knows before the forward goes into the status words here: [2] the path, [8 + 8l + 5] layer l's recurrence kernel,
// [8 + 8l + 6] the state slots its kernels run on (P, or P / 2 for a layer compacted to its live states).
struct StatusInit {
    int32_t path;
    int32_t rk[15];    // 8 + 8 * n_layers <= S5FXP_STATUS_WORDS
    int32_t slots[15];
    int32_t stream[15]; // [8 + 8l + 7]: state slots the recurrence streams hold (<= slots)
};
__global__ __launch_bounds__(256) void k_clear2(int32_t *a, int na, int32_t *b, int nb, StatusInit si, int n_layers, GroupOff go)
{
    gshift(a, (int64_t)blockIdx.y * go.status);
    gshift(b, (int64_t)blockIdx.y * go.ws);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < na + nb; i += gridDim.x * 256) {
        if (i < na) {
            int32_t v = 0;
            if (i == 2) v = si.path;
            else if (i >= 8 && (i & 7) == 5 && (i - 8) / 8 < n_layers) v = si.rk[(i - 8) / 8];
            else if (i >= 8 && (i & 7) == 6 && (i - 8) / 8 < n_layers) v = si.slots[(i - 8) / 8];
            else if (i >= 8 && (i & 7) == 7 && (i - 8) / 8 < n_layers) v = si.stream[(i - 8) / 8];
            a[i] = v;
        } else b[i - na] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// per-channel extremes of an int16 (N,H) tensor as POSITIVE floats: ext[h] = 65536 - min_h, ext[H+h] =
// 65536 + max_h (exact for 16-bit data).  Zero-initialised by a memset; atomicMax on the bit pattern;
// the multi-rank hook (element-wise float MAX) can exchange them as they are.
// ---------------------------------------------------------------------------------------------
constexpr float EXT_BIAS = 65536.f;
// Single-rank forwards spread the workgroups' atomics over EXT_REPS replicas of the 2H extremes (replica = blockIdx % EXT_REPS):
// 512 workgroups hitting the same 192 words serialise at the L2 atomic units for several microseconds, and that time sits
// on the critical path between two layers.  The finalize body folds the replicas with EXT_REPS independent loads.
constexpr int EXT_REPS = 8;

__device__ __forceinline__ void unpack8_i16(const v4i &w, int32_t (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = (int32_t)(int16_t)(w[i] & 0xffff);
        v[2 * i + 1] = w[i] >> 16;
    }
}

// block reduce of NV float maxima over blockDim.x threads (every thread gets the result)
template <int NV>
__device__ __forceinline__ void block_allmax(float (&v)[NV], float (*red)[8])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float t = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t = fmaxf(t, __shfl_xor(t, o, 64));
        if (lane == 0) red[i][wave] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float t = red[i][0];
        for (int w = 1; w < nw; ++w) t = fmaxf(t, red[i][w]);
        v[i] = t;
    }
    __syncthreads();
}

// One workgroup (256 threads >= H): thread h owns channel h and walks its two end points through the
// BatchNorm stages; between stages the workgroup reduces the maxima and thread 0 derives the exponent.
// Callable by any 256-thread workgroup: as its own kernel (k_bn_finalize_mm) or as the tail of the kernel that
// produced the extremes, run by the workgroup that finished last (k_resid_minmax16).  There the extremes were
// written by other workgroups' atomics, possibly on other XCDs: agent-scope loads, not cached ones.
// Returns the derived exponents to EVERY thread; `publish` chooses the workgroup that also writes them to *d and the
// status words (the B projection runs this in every workgroup's prologue: k_bproj_p).
__device__ __forceinline__ LayerDyn bn_finalize_mm_body(const BnArgs &a, const float *ext, int H, LayerDyn *d, int32_t *status,
                                                        int32_t *status_exps, int xe, int reps = 1, bool publish = true)
{
    // This runs on the critical path between two layers (nothing else is executing), so it is written for latency:
    // every operand a channel needs is requested up front, each stage costs ONE barrier (its own reduction slots),
    // and every thread derives the exponents itself from the reduced maxima instead of waiting for thread 0.
    __shared__ float red[4][3][8];
    const int h = threadIdx.x, lane = h & 63, wave = h >> 6, nw = (blockDim.x + 63) >> 6;
    const bool act = h < H;
    float e0 = 0.f, e1 = 0.f;
    int32_t pm = 0, pi = 0, ps = 0, pb = 0;
    if (act) {
        for (int rp = 0; rp < reps; ++rp) {
            e0 = fmaxf(e0, __hip_atomic_load(ext + rp * 2 * H + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            e1 = fmaxf(e1, __hip_atomic_load(ext + rp * 2 * H + H + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        pm = a.mm[h];
        pi = a.isv[h];
        if (a.scale) ps = a.scale[h];
        if (a.bias) pb = a.bias[h];
    }
    const int32_t xlo = act ? (int32_t)(EXT_BIAS - e0) : 0, xhi = act ? (int32_t)(e1 - EXT_BIAS) : 0;
    LayerDyn sd{};
    auto allmax = [&](int stage, float (&v)[3], int nv) {
        for (int i = 0; i < nv; ++i) {
            float t = v[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t = fmaxf(t, __shfl_xor(t, o, 64));
            if (lane == 0) red[stage][i][wave] = t;
        }
        __syncthreads();
        for (int i = 0; i < nv; ++i) {
            float t = red[stage][i][0];
            for (int w = 1; w < nw; ++w) t = fmaxf(t, red[stage][i][w]);
            v[i] = t;
        }
    };
    // the BatchNorm chain of s5fxp_kernels.hpp bn_chain<>, operands in registers
    auto c1 = [&](int32_t x) { return add_cb_apply(x, a.xb, pm, a.mb, sd.bn1, a.b1); };
    auto c2 = [&](int32_t x) { return sat(asr(wmul(c1(x), pi), sd.rs2), a.b2); };
    auto c3 = [&](int32_t x) {
        const int32_t t = c2(x);
        return a.scale ? sat(asr(wmul(t, ps), sd.rs3), a.b3) : t;
    };
    // ---- stage 1: x + (-mean)        fxpmodel.py:892-897
    {
        float v[3] = {0.f, 0.f, 0.f};
        if (act) {
            const float fm = tofloat(pm, a.me), f0 = tofloat(xlo, xe), f1 = tofloat(xhi, xe);
            v[0] = fmaxf(fabsf(__fadd_rn(f0, fm)), fabsf(__fadd_rn(f1, fm)));
        }
        allmax(0, v, 1); // only max |x + y| chooses the exponent (fxparray.py:421-425); the operands' own maxima size an
                         // intermediate width the device code does not need
        uint32_t m3[3] = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2])};
        sd.mx[0] = m3[0]; sd.mx[1] = m3[1]; sd.mx[2] = m3[2];
        sd.bn1 = finalize_add_cb(m3, xe, a.me, a.b1, status);
        sd.bn_e = sd.bn1.eo;
    }
    // ---- stage 2: * invsq_var        fxpmodel.py:902-907
    {
        float v[3] = {0.f, 0.f, 0.f};
        if (act) {
            const float fi = tofloat(pi, a.ie);
            const float f0 = tofloat(c1(xlo), sd.bn1.eo), f1 = tofloat(c1(xhi), sd.bn1.eo);
            v[0] = fmaxf(fabsf(__fmul_rn(f0, fi)), fabsf(__fmul_rn(f1, fi)));
        }
        allmax(1, v, 1);
        sd.mx[3] = __float_as_uint(v[0]);
        finalize_mul_cb(sd.mx[3], sd.bn1.eo, a.ie, a.b2, sd.rs2, sd.e2, status);
        sd.bn_e = sd.e2;
    }
    // ---- stage 3: * scale            fxpmodel.py:915-920
    if (a.scale) {
        float v[3] = {0.f, 0.f, 0.f};
        if (act) {
            const float fs = tofloat(ps, a.se);
            const float f0 = tofloat(c2(xlo), sd.e2), f1 = tofloat(c2(xhi), sd.e2);
            v[0] = fmaxf(fabsf(__fmul_rn(f0, fs)), fabsf(__fmul_rn(f1, fs)));
        }
        allmax(2, v, 1);
        sd.mx[4] = __float_as_uint(v[0]);
        finalize_mul_cb(sd.mx[4], sd.e2, a.se, a.b3, sd.rs3, sd.e3, status);
        sd.bn_e = sd.e3;
    }
    // ---- stage 4: + bias             fxpmodel.py:928-933
    if (a.bias) {
        const int e3 = a.scale ? sd.e3 : sd.e2;
        float v[3] = {0.f, 0.f, 0.f};
        if (act) {
            const float fb = tofloat(pb, a.be);
            const float f0 = tofloat(c3(xlo), e3), f1 = tofloat(c3(xhi), e3);
            v[0] = fmaxf(fabsf(__fadd_rn(f0, fb)), fabsf(__fadd_rn(f1, fb)));
        }
        allmax(3, v, 1);
        uint32_t m3[3] = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2])};
        sd.mx[5] = m3[0]; sd.mx[6] = m3[1]; sd.mx[7] = m3[2];
        sd.bn4 = finalize_add_cb(m3, e3, a.be, a.b4, status);
        sd.bn_e = sd.bn4.eo;
    }
    if (h == 0 && publish) { // (redo and the residual maxima slots of *d are written later in the layer)
        d->bn1 = sd.bn1; d->rs2 = sd.rs2; d->e2 = sd.e2; d->rs3 = sd.rs3; d->e3 = sd.e3; d->bn4 = sd.bn4; d->bn_e = sd.bn_e;
        for (int i = 0; i < 8; ++i) d->mx[i] = sd.mx[i];
        status_exps[0] = sd.bn1.eo;
        status_exps[1] = sd.e2;
        if (a.scale) status_exps[2] = sd.e3;
        if (a.bias) status_exps[3] = sd.bn4.eo;
    }
    return sd;
}

// stand-alone form (multi-rank mode: the extremes are exchanged between the two kernels)
__global__ __launch_bounds__(256) void k_bn_finalize_mm(BnArgs a, const float *ext, int H, LayerDyn *d, int32_t *status,
                                                        int32_t *status_exps)
{
    (void)bn_finalize_mm_body(a, ext, H, d, status, status_exps, a.xe.get());
}

// Residual add + ReLU of one layer and, in the same pass, the per-channel extremes of its result (the next
// layer's BatchNorm operand).  RESID=false: extremes of z only (the encoder output ahead of layer 0).
// block = 384 threads = G8 channel-groups (8 channels = 16 bytes each) x R frame lanes (32 at H=96, 16 at H=192);
// a workgroup owns `span` consecutive frames (a multiple of 4R) and keeps four frames per thread in flight.
// ext == nullptr: no extremes wanted (last layer).
// Two single-workgroup kernels are folded in (single-rank mode; with a multi-rank hook they stay separate
// because the ranks exchange maxima in between):
//   head  the residual add's compute_best exponent (k_res_finalize): every workgroup derives it from the three
//         maxima, workgroup 0 publishes it;
//   tail  the NEXT layer's BatchNorm exponents (k_bn_finalize_mm): the workgroup that takes the last ticket
//         runs it once all extremes are in.  No workgroup ever waits for another.
struct ResidHead {
    LayerDyn *d;
    int32_t res_exp;
    DynExp skip_e;
    int32_t redo_slot;
    int32_t *status_exps;
    int32_t enable; // 0: d->res was written by k_res_finalize
};

constexpr int RESID_THREADS = 384;
template <bool RESID>
__global__ __launch_bounds__(RESID_THREADS, 5) void k_resid_minmax16(const int16_t *z_, const int16_t *skip_, int16_t *out_, int32_t *tr_resid,
                                                                  int64_t N, int H, int64_t span, int res_bits, int skip_bits, ResidHead hd,
                                                                  float *ext, int ext_reps, int32_t *status, GroupOff go)
{
    // this group's tensors (scan_quad.hpp GroupOff); the three streams keep their no-alias promise
    const int64_t gws = (int64_t)blockIdx.y * go.ws, gst = (int64_t)blockIdx.y * go.status;
    const int16_t *__restrict__ z = reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(z_) + gws);
    const int16_t *__restrict__ skip = reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(skip_) + gws);
    int16_t *__restrict__ out = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(out_) + gws);
    gshift(ext, gws); gshift_nn(hd.d, gws); gshift(hd.skip_e.dyn, gws); gshift_nn(hd.status_exps, gst); gshift_nn(status, gst);
    __shared__ int32_t smin[RESID_THREADS * 8], smax[RESID_THREADS * 8];
    __shared__ AddCb sp;
    const int G = H >> 3, R = RESID_THREADS / G;
    const int g = threadIdx.x % G, rl = threadIdx.x / G;
    // a round = four frames per thread.  The first round's loads are issued BEFORE the head below: its exponent arithmetic
    // needs the maxima, the loads do not.  (Double-buffering every round was tried: 171 registers, one workgroup per CU
    // instead of three, 29 us instead of 23.)
    // frames of this workgroup: [lo_n, lo_n + cnt).  Everything inside is addressed as a wave-uniform base (the workgroup's
    // first frame) plus a 32-bit byte offset per thread: no 64-bit address arithmetic in the loop (the kernel sits at the
    // 96 registers that three workgroups per CU allow)
    const int64_t lo_n = (int64_t)blockIdx.x * span;
    const int cnt = (int)((lo_n + span < N ? lo_n + span : N) - lo_n);
    const char *zb = reinterpret_cast<const char *>(z + lo_n * H), *sb = reinterpret_cast<const char *>(skip + lo_n * H);
    char *ob = reinterpret_cast<char *>(out + lo_n * H);
    const unsigned rowb = 2u * (unsigned)H;                       // bytes per frame
    const unsigned toff = (unsigned)rl * rowb + 16u * (unsigned)g; // this thread's first vector
    const unsigned kstep = (unsigned)R * rowb;                    // R frames further
    auto fetch = [&](v4i(&zq)[4], v4i(&sq)[4], int i0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k * R + rl < cnt) {
                const unsigned o = toff + (unsigned)(i0 / R) * kstep + (unsigned)k * kstep;
                zq[k] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(zb + o));
                if constexpr (RESID) sq[k] = *reinterpret_cast<const v4i *>(sb + o);
            }
        }
    };
    v4i za[4], sa[4];
    const int first = rl, step = 4 * R; // frame indices relative to lo_n; i0 - rl is always a multiple of R
    if (first < cnt) fetch(za, sa, 0);
    AddCb p{};
    if constexpr (RESID) {
        if (hd.enable) {
            if (threadIdx.x == 0) {
                sp = finalize_add_cb(hd.d->mx + (hd.d->redo ? hd.redo_slot : 8), hd.res_exp, hd.skip_e.get(), res_bits, status);
                if (blockIdx.x == 0) {
                    hd.d->res = sp;
                    hd.status_exps[4] = sp.eo;
                }
            }
            __syncthreads();
            p = sp;
        } else {
            p = hd.d->res;
        }
    }
    AddCbV pv{};
    if constexpr (RESID) pv = make_add_cb_v(p, res_bits, skip_bits, res_bits);
    int32_t lo[8], hi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        lo[e] = 32767;
        hi[e] = -32768;
    }
    auto process = [&](const v4i(&zq)[4], const v4i(&sq)[4], int i0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k * R + rl < cnt) {
                int32_t v[8], s[8];
                unpack8_i16(zq[k], v);
                if constexpr (RESID) {
                    unpack8_i16(sq[k], s);
                    const unsigned o = toff + (unsigned)(i0 / R) * kstep + (unsigned)k * kstep;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int32_t rr = add_cb_apply(v[e], s[e], pv);
                        if (tr_resid) tr_resid[lo_n * H + (o >> 1) + e] = rr;
                        v[e] = rr < 0 ? 0 : rr;
                    }
                    const v2i a = pack4_i16(v[0], v[1], v[2], v[3]), b = pack4_i16(v[4], v[5], v[6], v[7]);
                    *reinterpret_cast<v4i *>(ob + o) = v4i{a[0], a[1], b[0], b[1]};
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    lo[e] = v[e] < lo[e] ? v[e] : lo[e];
                    hi[e] = v[e] > hi[e] ? v[e] : hi[e];
                }
            }
        }
    };
    for (int i0 = 0; i0 + rl < cnt; i0 += step) {
        process(za, sa, i0);
        if (i0 + step + rl < cnt) fetch(za, sa, i0 + step);
    }
    if (!ext) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        smin[threadIdx.x * 8 + e] = lo[e];
        smax[threadIdx.x * 8 + e] = hi[e];
    }
    __syncthreads();
    {
        // fold the R frame lanes: one thread per (bound, channel) -- two per item at H = 96, each folding half the lanes
        const int items = 2 * H, parts = RESID_THREADS / items > 1 ? 2 : 1, per = R / parts;
        const int item = threadIdx.x % items, part = threadIdx.x / items;
        const bool is_max = item >= H;
        const int c = is_max ? item - H : item;
        int32_t v = is_max ? -32768 : 32767;
        if (part < parts) {
            const int32_t *src = is_max ? smax : smin;
            for (int r = part * per; r < (part + 1) * per; ++r) {
                const int32_t t = src[r * G * 8 + c];
                v = is_max ? max(v, t) : min(v, t);
            }
        }
        __syncthreads();
        if (parts == 2 && part == 1) smin[item] = v; // the arrays are free now
        __syncthreads();
        if (part == 0) {
            if (parts == 2) {
                const int32_t t = smin[item];
                v = is_max ? max(v, t) : min(v, t);
            }
            const int rep = ext_reps > 1 ? (int)(blockIdx.x % ext_reps) : 0;
            atomicMax(reinterpret_cast<uint32_t *>(ext) + rep * 2 * H + item,
                      __float_as_uint(is_max ? EXT_BIAS + (float)v : EXT_BIAS - (float)v));
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Lean BatchNorm chain for the MFMA path.  Preconditions (host-checked, the same as for the extremes
// method): every BN operand <= 16 bit, exponents in [0,15]; the input is stored as int16 within its
// nominal bits.  Under them this is bn_chain<> with the data-independent parts moved out of the element:
//   * sat(v << 0, bits) == v for in-range v, so the "shift only if needed" selects disappear;
//   * "post > 0 ? << : >>" is (v << l) >> r with one of l, r zero;
//   * the per-channel operands (mean, bias) are shifted/saturated once per workgroup into LDS;
//   * 16 x 16-bit products fit v_mul_i32_i24 (full rate), whose low 32 bits are the exact product;
//   * change_cfg(t -> u) is a shift pair and one saturation at min(out_bits, u_bits).
// ---------------------------------------------------------------------------------------------
struct Bn16 {
    const int32_t *m1, *isv, *sc, *b4; // LDS, (H) each
    int32_t shx1, l1, r1, b1, xb;
    int32_t rs2, b2;
    int32_t has_sc, rs3, b3;
    int32_t has_b, shx4, tb4, l4, r4, b4b;
    int32_t cl, cr, cbits;
    fxp::SatB sxb, s1, s2, s3, stb4, s4b, scb; // the clip bounds of the chain, resident in VGPRs (fxp_prims.hpp sat_bounds)
};

// builds the LDS tables (all threads) and returns the scalars; call before a __syncthreads()
__device__ __forceinline__ Bn16 bn16_setup(const BnArgs &a, const LayerDyn &d, int32_t *lds, int H)
{
    Bn16 p;
    int32_t *m1 = lds, *isv = lds + H, *sc = lds + 2 * H, *b4 = lds + 3 * H;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        m1[h] = sat(wshl(a.mm[h], d.bn1.shy), a.mb);
        isv[h] = a.isv[h];
        sc[h] = a.scale ? a.scale[h] : 0;
        b4[h] = a.bias ? sat(wshl(a.bias[h], d.bn4.shy), a.bb) : 0;
    }
    p.m1 = m1; p.isv = isv; p.sc = sc; p.b4 = b4;
    p.shx1 = d.bn1.shx; p.l1 = d.bn1.post > 0 ? d.bn1.post : 0; p.r1 = d.bn1.post < 0 ? -d.bn1.post : 0;
    p.b1 = a.b1; p.xb = a.xb;
    p.rs2 = d.rs2; p.b2 = a.b2;
    p.has_sc = a.scale != nullptr; p.rs3 = d.rs3; p.b3 = a.b3;
    p.has_b = a.bias != nullptr; p.shx4 = d.bn4.shx; p.tb4 = a.scale ? a.b3 : a.b2;
    p.l4 = d.bn4.post > 0 ? d.bn4.post : 0; p.r4 = d.bn4.post < 0 ? -d.bn4.post : 0; p.b4b = a.b4;
    const int de = a.ue - d.bn_e;
    p.cl = de > 0 ? de : 0; p.cr = de < 0 ? -de : 0;
    p.cbits = a.out_bits < a.ub ? a.out_bits : a.ub;
    p.sxb = fxp::sat_bounds(p.xb); p.s1 = fxp::sat_bounds(p.b1); p.s2 = fxp::sat_bounds(p.b2);
    p.scb = fxp::sat_bounds(p.cbits);
    // (the scale / bias stages are rare: their bounds are only made when the stages exist)
    if (p.has_sc) p.s3 = fxp::sat_bounds(p.b3);
    if (p.has_b) { p.stb4 = fxp::sat_bounds(p.tb4); p.s4b = fxp::sat_bounds(p.b4b); }
    return p;
}

// four consecutive channels h0..h0+3: BatchNorm output t (pre_s5) and SSM input u
__device__ __forceinline__ void bn16_x4(const Bn16 &p, const int32_t (&x)[4], int h0, int32_t (&t)[4], int32_t (&u)[4])
{
    const v4i m4 = *reinterpret_cast<const v4i *>(p.m1 + h0), i4 = *reinterpret_cast<const v4i *>(p.isv + h0);
    v4i s4 = {0, 0, 0, 0}, b4 = {0, 0, 0, 0};
    if (p.has_sc) s4 = *reinterpret_cast<const v4i *>(p.sc + h0);
    if (p.has_b) b4 = *reinterpret_cast<const v4i *>(p.b4 + h0);
    // stage by stage over the four channels: the two optional stages cost one uniform branch per call, not one per element
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int32_t v = sat(asr(wshl(wadd(sat(wshl(x[e], p.shx1), p.sxb), m4[e]), p.l1), p.r1), p.s1);
        t[e] = sat(asr(__mul24(v, i4[e]), p.rs2), p.s2);
    }
    if (p.has_sc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = sat(asr(__mul24(t[e], s4[e]), p.rs3), p.s3);
    }
    if (p.has_b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = sat(asr(wshl(wadd(sat(wshl(t[e], p.shx4), p.stb4), b4[e]), p.l4), p.r4), p.s4b);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = sat(asr(wshl(t[e], p.cl), p.cr), p.scb);
}

// ---------------------------------------------------------------------------------------------
// B projection arguments (kernel: proj_p.hpp k_bproj_p)
// ---------------------------------------------------------------------------------------------
struct BprojM2Args {
    BnArgs bn;
    const int16_t *x; // (N,H)
    MfmaW w;
    int32_t *bq;      // native stream
    int16_t *u;       // (N,H) SSM input, for the C projection
    int32_t *tr_bu_re, *tr_bu_im, *tr_pre_s5, *tr_u; // traces (TRACE instantiation only)
    int64_t N;
    int32_t L, TB, H, P;
    int32_t rs_re, rs_im, bre_bits, bim_bits, sh_re, sh_im;
    int32_t t_lo, t_len; // k_bproj_p: the step range this launch covers (StepRange)
    int32_t k_re;        // SM = 2 (pair-native K stream): 2^16 - 2^(16 - A_re_exp), the addend of the negated product
    int32_t no_u;        // != 0: u is not stored -- the gate kernel recomputes it (mfma_fused.hpp k_cgate_p<.., GBN>)
    int32_t live_slots;  // SM = 3 / 1, > 0: only state slots below it are stored (scan_quad.hpp ScanPairLArgs::live_slots)
    // != nullptr: the per-channel extremes of the layer input (ext_reps replicas of 2H floats); every workgroup derives the
    // BatchNorm exponents from them in its prologue (bn_finalize_mm_body), workgroup 0 publishes them
    const float *ext;
    int32_t ext_reps;
    int32_t *status, *status_exps;
};

} // namespace s5
