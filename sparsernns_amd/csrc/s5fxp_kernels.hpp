// s5fxp_kernels.hpp -- HIP kernels of the fixed-point S5 forward (gfx950 / MI355X).
//
// Frame-tiled integer projections with fused prologues/epilogues, the exact sequential
// diagonal-SSM recurrence, and the data-dependent-exponent ("compute_best") reductions.
// Reference semantics: see fxp_prims.hpp and the per-kernel citations below
// (paths into /root/reference/sparseRNNs/).
#pragma once
#include "fxp_prims.hpp"
#include "scan_quad.hpp"

namespace s5 {
using namespace fxp;

// An exponent that is either known on the host or chosen on the device by an earlier
// compute_best op of the same forward.
struct DynExp {
    int32_t stat;
    const int32_t *dyn;
    __device__ __forceinline__ int get() const { return dyn ? *dyn : stat; }
};

// Per-layer device state written by the finalize kernels.
struct LayerDyn {
    AddCb bn1;         // x + (-mean)                    fxpmodel.py:892-897
    int32_t rs2, e2;   // * invsq_var                    fxpmodel.py:902-907
    int32_t rs3, e3;   // * scale                        fxpmodel.py:915-920
    AddCb bn4;         // + bias                         fxpmodel.py:928-933
    int32_t bn_e;      // exponent of the BatchNorm output
    int32_t pad0;
    AddCb res;         // gate + skip                    fxpmodel.py:1147-1152
    uint32_t mx[16];   // float32 maxima as bit patterns: [0..2] bn1, [3] bn2, [4] bn3, [5..7] bn4, [8..10] res
    int32_t redo;      // 1 if a state exceeded the fast kernels' exactness bound: the exact kernels re-run the layer
    int32_t pad1[3];
};

struct BnArgs {
    const int32_t *mm, *isv, *scale, *bias; // (H) each; scale/bias nullable
    int32_t xb;                             // layer-input bits
    DynExp xe;                              // layer-input exponent
    int32_t mb, me, b1;
    int32_t ib, ie, b2;
    int32_t sb, se, b3;
    int32_t bb, be, b4;
    int32_t ub, ue;      // SSM input config, fxpmodel.py:620-624
    int32_t out_bits;    // bits of the BatchNorm output
    const LayerDyn *dyn;
};

// BatchNorm chain up to and including stage UPTO (1..4); stage 5 = change_cfg to the SSM input.
template <int UPTO>
__device__ __forceinline__ int32_t bn_chain(const BnArgs &a, const LayerDyn &d, int32_t x, int h)
{
    int32_t t = add_cb_apply(x, a.xb, a.mm[h], a.mb, d.bn1, a.b1);
    if (UPTO == 1) return t;
    t = sat(asr(wmul(t, a.isv[h]), d.rs2), a.b2);
    if (UPTO == 2) return t;
    if (a.scale) t = sat(asr(wmul(t, a.scale[h]), d.rs3), a.b3);
    if (UPTO == 3) return t;
    if (a.bias) t = add_cb_apply(t, a.scale ? a.b3 : a.b2, a.bias[h], a.bb, d.bn4, a.b4);
    if (UPTO == 4) return t;
    return chcfg(t, a.out_bits, d.bn_e, a.ub, a.ue);
}

// ---------------------------------------------------------------------------------------------
// block-level float max -> global atomicMax on the bit pattern (values are >= 0)
// ---------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_max_atomic(float (&v)[NV], uint32_t *dst)
{
    __shared__ float red[NV][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
        if (lane == 0) red[i][wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float x = red[threadIdx.x][0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) x = fmaxf(x, red[threadIdx.x][w]);
        atomicMax(dst + threadIdx.x, __float_as_uint(x));
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm reductions: the float32 maxima each compute_best op needs (fxparray.py:421-430,602-607)
// ---------------------------------------------------------------------------------------------
template <int STAGE>
__global__ __launch_bounds__(256) void k_bn_reduce(BnArgs a, const int32_t *__restrict__ x, int64_t NH, int H,
                                                   LayerDyn *dynw)
{
    const LayerDyn d = *a.dyn;
    const int xe = a.xe.get();
    float v[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NH; i += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(i % H);
        const int32_t xv = x[i];
        if (STAGE == 1) {
            const float fx = tofloat(xv, xe), fm = tofloat(a.mm[h], a.me);
            v[0] = fmaxf(v[0], fabsf(__fadd_rn(fx, fm)));
            v[1] = fmaxf(v[1], fabsf(fx));
            v[2] = fmaxf(v[2], fabsf(fm));
        } else if (STAGE == 2) {
            const int32_t t = bn_chain<1>(a, d, xv, h);
            v[0] = fmaxf(v[0], fabsf(__fmul_rn(tofloat(t, d.bn1.eo), tofloat(a.isv[h], a.ie))));
        } else if (STAGE == 3) {
            const int32_t t = bn_chain<2>(a, d, xv, h);
            v[0] = fmaxf(v[0], fabsf(__fmul_rn(tofloat(t, d.e2), tofloat(a.scale[h], a.se))));
        } else {
            const int32_t t = bn_chain<3>(a, d, xv, h);
            const float ft = tofloat(t, a.scale ? d.e3 : d.e2), fb = tofloat(a.bias[h], a.be);
            v[0] = fmaxf(v[0], fabsf(__fadd_rn(ft, fb)));
            v[1] = fmaxf(v[1], fabsf(ft));
            v[2] = fmaxf(v[2], fabsf(fb));
        }
    }
    constexpr int slot = STAGE == 1 ? 0 : (STAGE == 2 ? 3 : (STAGE == 3 ? 4 : 5));
    if (STAGE == 1 || STAGE == 4) block_max_atomic<3>(v, dynw->mx + slot);
    else {
        float w[1] = {v[0]};
        block_max_atomic<1>(w, dynw->mx + slot);
    }
}

enum { ST_NEGSHIFT = 1, ST_NEGEXP = 2, ST_WIDE_STATE = 4, ST_WIDE_INPUT = 8, ST_REDO = 16 };

__device__ inline AddCb finalize_add_cb(const uint32_t *mx, int xe, int ye, int ob, int32_t *status)
{
    const int ib = intbits_f32(__uint_as_float(mx[0]), 1e-6f);
    AddCb p;
    p.eo = ob - ib - 1;
    const int ea = xe > ye ? xe : ye;
    p.shx = ea - xe;
    p.shy = ea - ye;
    p.post = p.eo - ea;
    if (p.eo < 0) atomicOr(status, ST_NEGEXP);
    if (p.post > 31 || p.post < -31 || p.shx > 31 || p.shy > 31) {
        atomicOr(status, ST_NEGSHIFT);
        p.post = p.post > 31 ? 31 : (p.post < -31 ? -31 : p.post);
        p.shx = p.shx > 31 ? 31 : p.shx;
        p.shy = p.shy > 31 ? 31 : p.shy;
    }
    return p;
}

__device__ inline void finalize_mul_cb(uint32_t mx, int xe, int ye, int ob, int32_t &rs, int32_t &eo, int32_t *status)
{
    eo = ob - intbits_f32(__uint_as_float(mx), 1e-6f) - 1;
    rs = xe + ye - eo;
    if (eo < 0) atomicOr(status, ST_NEGEXP);
    if (rs < 0 || rs > 31) { // fxparray.py:619-621 raises ValueError for rs < 0
        atomicOr(status, ST_NEGSHIFT);
        rs = rs < 0 ? 0 : 31;
    }
}

// One thread turns the maxima of BatchNorm stage STAGE into shifts/exponents.
template <int STAGE>
__global__ void k_bn_finalize(BnArgs a, LayerDyn *d, int32_t *status, int32_t *status_exps)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (STAGE == 1) {
        d->bn1 = finalize_add_cb(d->mx + 0, a.xe.get(), a.me, a.b1, status);
        status_exps[0] = d->bn1.eo;
        d->bn_e = d->bn1.eo;
    } else if (STAGE == 2) {
        finalize_mul_cb(d->mx[3], d->bn1.eo, a.ie, a.b2, d->rs2, d->e2, status);
        status_exps[1] = d->e2;
        d->bn_e = d->e2;
    } else if (STAGE == 3) {
        finalize_mul_cb(d->mx[4], d->e2, a.se, a.b3, d->rs3, d->e3, status);
        status_exps[2] = d->e3;
        d->bn_e = d->e3;
    } else {
        d->bn4 = finalize_add_cb(d->mx + 5, a.scale ? d->e3 : d->e2, a.be, a.b4, status);
        status_exps[3] = d->bn4.eo;
        d->bn_e = d->bn4.eo;
    }
}

__global__ void k_res_finalize(LayerDyn *d, int res_exp, DynExp skip_e, int res_bits, int32_t *status,
                               int32_t *status_exps, int redo_slot)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // MFMA path: the exact re-run of a layer (redo) leaves its maxima in slots 11..13 (redo_slot), the fused
    // fast kernel's are in 8..10; the generic path has a single gate kernel and passes redo_slot = 8
    d->res = finalize_add_cb(d->mx + (d->redo ? redo_slot : 8), res_exp, skip_e.get(), res_bits, status);
    status_exps[4] = d->res.eo;
}

// ---------------------------------------------------------------------------------------------
// Frame-tiled integer matmul core.  A block owns 64 consecutive frames; lane = frame, the four
// waves split the output columns (mw columns each, mw % 4 == 0).  x is staged [frame][k] (padded,
// conflict-free column reads), the weight chunk [k][col] is read as broadcast 16-byte rows.
// X24: both operands are known to fit 24 signed bits -> full-rate v_mad_i32_i24, whose low 32
// bits equal the int32 product; otherwise 32-bit multiplies.  Either way the accumulation wraps
// modulo 2^32 exactly like fxparray.py:662.
// ---------------------------------------------------------------------------------------------
constexpr int TN = 64; // frames per block
constexpr int KC = 32; // k chunk

template <int MWMAX>
struct MMShared {
    union {
        struct {
            int32_t xs[TN][KC + 1];
            int32_t ws[KC][4 * MWMAX];
        } in;
        int32_t out[TN][4 * MWMAX + 1];
    };
};

template <bool X24>
__device__ __forceinline__ int32_t mac(int32_t acc, int32_t x, int32_t w)
{
    if (X24) return wadd(acc, __mul24(x, w));
    return wadd(acc, wmul(x, w));
}

template <int MWMAX, bool X24, class LoadX>
__device__ __forceinline__ void mm_accumulate(MMShared<MWMAX> &S, int32_t (&acc)[MWMAX], LoadX loadx,
                                              const int32_t *__restrict__ W, int K, int M, int mw, int64_t n0,
                                              int64_t N)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int j = 0; j < MWMAX; ++j) acc[j] = 0;
    for (int k0 = 0; k0 < K; k0 += KC) {
#pragma unroll
        for (int i = 0; i < (TN * KC) / 256; ++i) {
            const int e = tid + 256 * i, nl = e / KC, kl = e % KC;
            const int k = k0 + kl;
            const int64_t n = n0 + nl;
            S.in.xs[nl][kl] = (k < K && n < N) ? loadx(n, k) : 0;
        }
        for (int e = tid; e < KC * 4 * MWMAX; e += 256) {
            const int kl = e / (4 * MWMAX), c = e % (4 * MWMAX);
            const int k = k0 + kl;
            S.in.ws[kl][c] = (k < K && c < M) ? W[(int64_t)k * M + c] : 0;
        }
        __syncthreads();
#pragma unroll 4
        for (int kl = 0; kl < KC; ++kl) {
            const int32_t xv = S.in.xs[lane][kl];
#pragma unroll
            for (int j = 0; j < MWMAX; j += 4) {
                if (j < mw) {
                    const int4 w4 = *reinterpret_cast<const int4 *>(&S.in.ws[kl][wave * mw + j]);
                    acc[j + 0] = mac<X24>(acc[j + 0], xv, w4.x);
                    acc[j + 1] = mac<X24>(acc[j + 1], xv, w4.y);
                    acc[j + 2] = mac<X24>(acc[j + 2], xv, w4.z);
                    acc[j + 3] = mac<X24>(acc[j + 3], xv, w4.w);
                }
            }
        }
        __syncthreads();
    }
}

// accumulators -> LDS [frame][col] so that the epilogue runs with lanes over columns (coalesced)
template <int MWMAX>
__device__ __forceinline__ void mm_stage_out(int32_t (*out)[4 * MWMAX + 1], const int32_t (&acc)[MWMAX], int mw)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < MWMAX; ++j)
        if (j < mw) out[lane][wave * mw + j] = acc[j];
}

// ---------------------------------------------------------------------------------------------
// Pruned dense layer: fxp_matmul (+ bias, ReLU) with the (K,M) weight stored by output channel -- CSR of the
// transposed kernel: rowptr[M+1], colidx (the k of every non-zero), val.  int32 sums wrap modulo 2^32 whatever the
// order of the terms, so skipping the zeros changes nothing (fxparray.py:662).
// A workgroup owns 64 frames: the whole (64,K) input tile goes to LDS once (row stride odd: conflict-free column
// reads); a wave takes one output channel at a time, lane = frame, and walks that channel's non-zeros -- the
// (k, value) pairs are wave-uniform, so they come through the scalar cache.  Results are staged [frame][64 channels]
// and stored with lanes over channels.  LDS: [x 64 x (K|1)] [out 64 x 65].
// On MI355X this is for operands the int8 MFMA path cannot take (activations beyond 16 bit, arbitrary shapes): there
// the weights live in registers and the dense contraction costs ~2 us per kernel, which no sparse format can beat.
// ---------------------------------------------------------------------------------------------
struct DenseCsrArgs {
    const int32_t *x;                       // (N,K)
    const int32_t *rowptr, *colidx, *val;   // by output channel
    const int32_t *bias;                    // (M) or null
    int32_t *y;                             // (N,M)
    int64_t N;
    int32_t K, M;
    int32_t rs, b_bits, b_exp, out_bits, out_exp, relu;
};

__global__ __launch_bounds__(256) void k_dense_csr(DenseCsrArgs a)
{
    extern __shared__ int32_t csr_smem[];
    const int KS = a.K | 1;
    int32_t *xs = csr_smem, *out = csr_smem + 64 * KS;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int e = threadIdx.x; e < 64 * a.K; e += 256) {
        const int nl = e / a.K, k = e - nl * a.K;
        const int64_t n = n0 + nl;
        xs[nl * KS + k] = n < a.N ? a.x[n * a.K + k] : 0;
    }
    __syncthreads();
    for (int m0 = 0; m0 < a.M; m0 += 64) {
        for (int mj = wave; mj < 64 && m0 + mj < a.M; mj += 4) {
            const int m = m0 + mj;
            const int t0 = a.rowptr[m], t1 = a.rowptr[m + 1];
            int32_t acc = 0;
            for (int t = t0; t < t1; ++t) acc = wadd(acc, wmul(xs[lane * KS + a.colidx[t]], a.val[t]));
            out[lane * 65 + mj] = acc;
        }
        __syncthreads();
        for (int nl = wave; nl < 64; nl += 4) {
            const int64_t n = n0 + nl;
            const int m = m0 + lane;
            if (n < a.N && m < a.M) {
                int32_t v = sat(asr(out[nl * 65 + lane], a.rs), a.out_bits);
                if (a.bias) v = sat(wadd(v, chexp(a.bias[m], a.b_bits, a.b_exp, a.out_exp)), a.out_bits);
                if (a.relu) v = v < 0 ? 0 : v;
                a.y[n * a.M + m] = v;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Dense: FxpDense.forward (fxpmodel.py:331-366) [+ ReLU fxpmodel.py:53-63].
// ---------------------------------------------------------------------------------------------
struct DenseArgs {
    const int32_t *x;    // (N,K)
    const int32_t *w;    // (K,M)
    const int32_t *bias; // (M) or null
    int32_t *y;          // (N,M)
    int64_t N;
    int32_t K, M, mw;
    int32_t xb;
    DynExp xe;
    int32_t inp_bits, inp_exp; // the conversion of fxpmodel.py:335-347 is applied when check_inp != 0
    int32_t check_inp;
    int32_t w_exp, b_bits, b_exp, out_bits, out_exp;
    int32_t relu;
    int32_t check24; // flag inputs that do not fit 24 bits (X24 kernels only)
    int32_t *status;
};

template <int MWMAX, bool X24>
__global__ __launch_bounds__(256) void k_dense(DenseArgs a)
{
    __shared__ MMShared<MWMAX> S;
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const int xe0 = a.xe.get();
    const bool conv = a.check_inp && (a.xb > a.inp_bits || xe0 > a.inp_exp);
    const int xe = conv ? a.inp_exp : xe0;
    int rs = xe + a.w_exp - a.out_exp;
    if (rs < 0 || rs > 31) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(a.status, ST_NEGSHIFT);
        rs = rs < 0 ? 0 : 31;
    }
    int32_t acc[MWMAX];
    bool wide = false;
    auto loadx = [&](int64_t n, int k) {
        int32_t v = a.x[n * a.K + k];
        if (conv) v = chcfg(v, a.xb, xe0, a.inp_bits, a.inp_exp);
        if (X24) wide |= (v != asr(wshl(v, 8), 8));
        return v;
    };
    mm_accumulate<MWMAX, X24>(S, acc, loadx, a.w, a.K, a.M, a.mw, n0, a.N);
    if (X24 && a.check24 && __any(wide) && (threadIdx.x & 63) == 0) atomicOr(a.status, ST_WIDE_INPUT);
    mm_stage_out<MWMAX>(S.out, acc, a.mw);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int nl = wave; nl < TN; nl += 4) {
        const int64_t n = n0 + nl;
        if (n >= a.N) break;
        for (int m = lane; m < a.M; m += 64) {
            int32_t v = sat(asr(S.out[nl][m], rs), a.out_bits);
            if (a.bias) v = sat(wadd(v, chexp(a.bias[m], a.b_bits, a.b_exp, a.out_exp)), a.out_bits);
            if (a.relu) v = v < 0 ? 0 : v;
            a.y[n * a.M + m] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// B projection: BatchNorm chain + change_cfg -> u, Bu = u @ [B_re^T | B_im^T]
// (fxpmodel.py:620-644).  W is (H, 2P): column p = B_re[p][:], column P+p = B_im[p][:].
// ---------------------------------------------------------------------------------------------
struct BprojArgs {
    BnArgs bn;
    const int32_t *x; // (N,H) layer input
    const int32_t *w; // (H,2P)
    int32_t *bq;               // native stream: Bu shifted to the state exponent
    int32_t *tr_bu_re, *tr_bu_im; // optional traces (N,P): Bu as the reference stores it
    int32_t *tr_pre_s5, *tr_u; // optional traces (N,H)
    int64_t N;
    int32_t L, TB;
    int32_t H, P, mw;
    int32_t rs_re, rs_im, bre_bits, bim_bits;
    int32_t sh_re, sh_im;      // Bu exponent - x exponent
};

template <int MWMAX, bool X24>
__global__ __launch_bounds__(256) void k_bproj(BprojArgs a)
{
    __shared__ MMShared<MWMAX> S;
    const LayerDyn d = *a.bn.dyn;
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    int32_t acc[MWMAX];
    auto loadx = [&](int64_t n, int k) {
        const int32_t xv = a.x[n * a.H + k];
        const int32_t t = bn_chain<4>(a.bn, d, xv, k);
        const int32_t u = chcfg(t, a.bn.out_bits, d.bn_e, a.bn.ub, a.bn.ue);
        if (a.tr_pre_s5) a.tr_pre_s5[n * a.H + k] = t;
        if (a.tr_u) a.tr_u[n * a.H + k] = u;
        return u;
    };
    mm_accumulate<MWMAX, X24>(S, acc, loadx, a.w, a.H, 2 * a.P, a.mw, n0, a.N);
    mm_stage_out<MWMAX>(S.out, acc, a.mw);
    __syncthreads();
    // epilogue in stream order [time block][state][re|im][step]: consecutive threads write consecutive words
    const int per_blk = a.P * 8;
    for (int idx = threadIdx.x; idx < (TN / 4) * per_blk; idx += 256) {
        const int tbl = idx / per_blk, rem = idx - tbl * per_blk;
        const int p = rem >> 3, c = (rem >> 2) & 1, j = rem & 3;
        const int nl = tbl * 4 + j;
        const int64_t n = n0 + nl;
        if (n >= a.N) continue;
        const int32_t raw = S.out[nl][c * a.P + p];
        // Bu = sat(asr(u@B, rs)) (fxparray.py:667-676), then the scan's shiftto (fxpmodel.py:158-167)
        const int32_t bu = c ? sat(asr(raw, a.rs_im), a.bim_bits) : sat(asr(raw, a.rs_re), a.bre_bits);
        const int sh = c ? a.sh_im : a.sh_re;
        const int64_t b = n / a.L;
        const int t = (int)(n - b * a.L);
        a.bq[native_word(b, t, p, c, a.TB, a.P)] = sh > 0 ? asr(bu, sh) : wshl(bu, -sh);
        if (c == 0 && a.tr_bu_re) a.tr_bu_re[n * a.P + p] = bu;
        if (c == 1 && a.tr_bu_im) a.tr_bu_im[n * a.P + p] = bu;
    }
}

// ---------------------------------------------------------------------------------------------
// "Scan-native" stream layout shared by the B projection (writer), the recurrence kernels and the
// C projection (reader); see scan_quad.hpp.  TB = time blocks of 4 steps per sequence (padded).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// The sequential recurrence (fxpmodel.py:147-208), one lane per (sequence, state), 32-bit
// multiplies: exact for every int32 input.  Used by the op-level s5fxp_scan (plain (B,L,P) layout)
// and as the fallback of the model forward (native layout) when the fast kernel's range check fails.
//   re' = asr(Ar*xr, eAr) - asr(Ai*xi, eAr) + shiftto(Bu_re)
//   im' = asr(Ar*xi, eAi) + asr(Ai*xr, eAi) + shiftto(Bu_im)          no clip, int32 wrap.
// ---------------------------------------------------------------------------------------------
struct ScanArgs {
    const int32_t *bu_re, *bu_im; // plain: (B,L,P) each.  native: bu_re = stream (already shifted), bu_im unused
    const int32_t *a_re, *a_im;   // (P)
    int32_t *out_re, *out_im;     // plain: (B,L,P) each (post-ReLU if relu).  native: out_re = stream of raw states
    int32_t B, L, P, TB;
    int32_t ea_re, ea_im;
    int32_t sh_re, sh_im; // Bu exponent - x exponent: > 0 right shift, <= 0 left shift (fxpmodel.py:158-167)
    int32_t relu;
    const int32_t *run_if; // native fallback: run only when *run_if != 0 (nullptr: always)
    const int32_t *x0_re, *x0_im; // native: (B,P) state before the first step (streaming carry), nullptr = zeros
};

__device__ __forceinline__ void scan_step(int32_t Ar, int32_t Ai, int ea_re, int ea_im, int32_t br, int32_t bi,
                                          int32_t &xr, int32_t &xi)
{
    const int32_t rr = wadd(wsub(asr(wmul(Ar, xr), ea_re), asr(wmul(Ai, xi), ea_re)), br);
    const int32_t ri = wadd(wadd(asr(wmul(Ar, xi), ea_im), asr(wmul(Ai, xr), ea_im)), bi);
    xr = rr;
    xi = ri;
}

template <int UNROLL>
__global__ __launch_bounds__(64) void k_scan_lane(ScanArgs a)
{
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.P;
    const bool active = gid < total;
    const int64_t g = active ? gid : total - 1;
    const int b = (int)(g / a.P), p = (int)(g % a.P);
    const int64_t base = (int64_t)b * a.L * a.P + p;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    int32_t xr = 0, xi = 0;
    int32_t r0[UNROLL], i0[UNROLL], r1[UNROLL], i1[UNROLL];
    auto load = [&](int32_t (&r)[UNROLL], int32_t (&i)[UNROLL], int t0) {
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int t = t0 + j;
            const int64_t off = base + (int64_t)(t < a.L ? t : a.L - 1) * a.P;
            r[j] = a.bu_re[off];
            i[j] = a.bu_im[off];
        }
    };
    auto run = [&](const int32_t (&r)[UNROLL], const int32_t (&i)[UNROLL], int t0) {
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            const int t = t0 + j;
            if (t < a.L) {
                const int32_t br = a.sh_re > 0 ? asr(r[j], a.sh_re) : wshl(r[j], -a.sh_re);
                const int32_t bi = a.sh_im > 0 ? asr(i[j], a.sh_im) : wshl(i[j], -a.sh_im);
                scan_step(Ar, Ai, a.ea_re, a.ea_im, br, bi, xr, xi);
                if (active) {
                    int32_t sr = xr, si = xi;
                    if (a.relu) crelu(sr, si);
                    const int64_t off = base + (int64_t)t * a.P;
                    a.out_re[off] = sr;
                    a.out_im[off] = si;
                }
            }
        }
    };
    // two register sets alternate, so a prefetched chunk is never copied (a copy would force vmcnt(0))
    load(r0, i0, 0);
    for (int t0 = 0; t0 < a.L; t0 += 2 * UNROLL) {
        load(r1, i1, t0 + UNROLL);
        run(r0, i0, t0);
        load(r0, i0, t0 + 2 * UNROLL);
        run(r1, i1, t0 + UNROLL);
    }
}

// native-layout variant: 4 steps per 16-byte access
__global__ __launch_bounds__(64) void k_scan_lane_native(ScanArgs a)
{
    if (a.run_if && *a.run_if == 0) return;
    const int64_t gid = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t total = (int64_t)a.B * a.P;
    if (gid >= total) return;
    const int b = (int)(gid / a.P), p = (int)(gid % a.P);
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int nblk = (a.L + 3) >> 2;
    int32_t xr = a.x0_re ? a.x0_re[gid] : 0, xi = a.x0_re ? a.x0_im[gid] : 0;
    for (int tb = 0; tb < nblk; ++tb) {
        const int64_t w = native_word(b, tb << 2, p, 0, a.TB, a.P);
        const int4 vr = *reinterpret_cast<const int4 *>(a.bu_re + w);
        const int4 vi = *reinterpret_cast<const int4 *>(a.bu_re + w + 4);
        int4 orr, oi;
        scan_step(Ar, Ai, a.ea_re, a.ea_im, vr.x, vi.x, xr, xi); orr.x = xr; oi.x = xi;
        scan_step(Ar, Ai, a.ea_re, a.ea_im, vr.y, vi.y, xr, xi); orr.y = xr; oi.y = xi;
        scan_step(Ar, Ai, a.ea_re, a.ea_im, vr.z, vi.z, xr, xi); orr.z = xr; oi.z = xi;
        scan_step(Ar, Ai, a.ea_re, a.ea_im, vr.w, vi.w, xr, xi); orr.w = xr; oi.w = xi;
        *reinterpret_cast<int4 *>(a.out_re + w) = orr;
        *reinterpret_cast<int4 *>(a.out_re + w + 4) = oi;
    }
}

// native stream -> plain (B,L,P) pair (traces only)
__global__ void k_unpack_native(const int32_t *__restrict__ stream, int32_t *__restrict__ re, int32_t *__restrict__ im,
                                int B, int L, int P, int TB)
{
    const int64_t n = (int64_t)B * L * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % P);
        const int64_t bt = i / P;
        const int t = (int)(bt % L);
        const int64_t b = bt / L;
        const int64_t w = native_word(b, t, p, 0, TB, P);
        if (re) re[i] = stream[w];
        if (im) im[i] = stream[w + 4];
    }
}

// ---------------------------------------------------------------------------------------------
// C projection + feed-through (fxpmodel.py:746-793) + ReLU (fxpmodel.py:1125):
//   cx = sat(sat(asr(xr@C_re^T)) - sat(asr(xi@C_im^T))); y = sat(2*cx + sat(asr(D*u)))
// Reads the RAW states from the native stream and applies the complex ReLU (fxpmodel.py:740-742)
// while loading.  W_re / W_im are (P,H) = C_re^T / C_im^T.  u is recomputed from the layer input.
// PASS 0 (fast): also range-checks every raw state against xmax (the exactness bound of the
//   recurrence kernel that produced it, and < 2^23 so the 24-bit multiplies here are exact); a
//   violation sets LayerDyn::redo.  PASS 1 (exact): runs only when redo is set, 32-bit multiplies.
// ---------------------------------------------------------------------------------------------
struct CprojArgs {
    BnArgs bn;
    const void *x;             // (N,H) layer input (for u), element type AT
    const int32_t *xs;         // native stream of raw states
    const int32_t *w_re, *w_im; // (P,H)
    const int32_t *D;          // (H)
    void *x1;                  // (N,H) relu(ys), element type AT
    int32_t *tr_ys;            // optional (N,H)
    int64_t N;
    int32_t L, TB;
    int32_t H, P, mw;
    int32_t rs_re, rs_im, rs_d, y_bits;
    int32_t xmax;
    LayerDyn *dynw;
    int32_t *status;
};

// AT = storage type of the (N,H) activations: int32_t (generic path) or int16_t (MFMA path's fallback)
template <int MWMAX, bool X24, int PASS, typename AT>
__global__ __launch_bounds__(256) void k_cproj(CprojArgs a)
{
    __shared__ MMShared<MWMAX> S;
    __shared__ int32_t out_re[TN][4 * MWMAX + 1];
    const LayerDyn d = *a.bn.dyn;
    if (PASS == 1 && d.redo == 0) return;
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    int32_t acc[MWMAX];
    bool bad = false;
    auto load_c = [&](int64_t n, int k, int c) {
        const int64_t b = n / a.L;
        const int t = (int)(n - b * a.L);
        const int64_t w = native_word(b, t, k, 0, a.TB, a.P);
        int32_t re = a.xs[w], im = a.xs[w + 4];
        if (PASS == 0) bad |= (re > a.xmax) | (re < -a.xmax) | (im > a.xmax) | (im < -a.xmax);
        crelu(re, im);
        return c ? im : re;
    };
    mm_accumulate<MWMAX, X24>(S, acc, [&](int64_t n, int k) { return load_c(n, k, 0); }, a.w_re, a.P, a.H, a.mw, n0, a.N);
    mm_stage_out<MWMAX>(out_re, acc, a.mw);
    mm_accumulate<MWMAX, X24>(S, acc, [&](int64_t n, int k) { return load_c(n, k, 1); }, a.w_im, a.P, a.H, a.mw, n0, a.N);
    mm_stage_out<MWMAX>(S.out, acc, a.mw);
    if (PASS == 0 && __any(bad) && (threadIdx.x & 63) == 0) {
        atomicExch(&a.dynw->redo, 1);
        atomicOr(a.status, ST_WIDE_STATE);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int nl = wave; nl < TN; nl += 4) {
        const int64_t n = n0 + nl;
        if (n >= a.N) break;
        for (int m = lane; m < a.H; m += 64) {
            const int32_t cr = sat(asr(out_re[nl][m], a.rs_re), a.y_bits);
            const int32_t ci = sat(asr(S.out[nl][m], a.rs_im), a.y_bits);
            const int32_t cx = sat(wadd(cr, wmul(ci, -1)), a.y_bits);
            const int32_t cx2 = wmul(cx, 2); // not clipped, fxpmodel.py:765-767
            const int32_t u = bn_chain<5>(a.bn, d, (int32_t) reinterpret_cast<const AT *>(a.x)[n * a.H + m], m);
            const int32_t du = sat(asr(wmul(a.D[m], u), a.rs_d), a.y_bits);
            const int32_t y = sat(wadd(cx2, du), a.y_bits);
            if (a.tr_ys) a.tr_ys[n * a.H + m] = y;
            reinterpret_cast<AT *>(a.x1)[n * a.H + m] = (AT)(y < 0 ? 0 : y);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// out2 dense + LUT sigmoid + mult_gate (fxpmodel.py:1133-1137, 97-144, 1075-1093), plus the
// float32 maxima of the residual compute_best add (fxpmodel.py:1147-1152).
// ---------------------------------------------------------------------------------------------
struct GateArgs {
    const int32_t *x1;   // (N,H) relu(ys), bits y_bits, exp y_exp
    const int32_t *w;    // (H,H)
    const int32_t *bias; // (H)
    const int32_t *skip; // (N,H) layer input
    int32_t *z;          // (N,H)
    int32_t *tr_out2, *tr_sig; // optional
    int64_t N;
    int32_t H, mw;
    int32_t y_bits, y_exp;
    int32_t inp_bits, inp_exp, w_exp, b_bits, b_exp, out_bits, out_exp;
    int32_t sig_x, sig_y;
    int32_t lut[8];
    int32_t l_bits, l_exp, r_bits, r_exp, res_bits, res_exp, rs_gate;
    int32_t skip_bits;
    DynExp skip_e;
    LayerDyn *dynw;
    int32_t *status;
};

template <int MWMAX, bool X24>
__global__ __launch_bounds__(256) void k_out2gate(GateArgs a)
{
    __shared__ MMShared<MWMAX> S;
    __shared__ int32_t lut[8];
    if (threadIdx.x < 8) lut[threadIdx.x] = a.lut[threadIdx.x];
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const bool conv = (a.y_bits > a.inp_bits) || (a.y_exp > a.inp_exp);
    const int xe = conv ? a.inp_exp : a.y_exp;
    int rs = xe + a.w_exp - a.out_exp;
    if (rs < 0 || rs > 31) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(a.status, ST_NEGSHIFT);
        rs = rs < 0 ? 0 : 31;
    }
    const int skip_e = a.skip_e.get();
    int32_t acc[MWMAX];
    auto loadx = [&](int64_t n, int k) {
        const int32_t v = a.x1[n * a.H + k];
        return conv ? chcfg(v, a.y_bits, a.y_exp, a.inp_bits, a.inp_exp) : v;
    };
    mm_accumulate<MWMAX, X24>(S, acc, loadx, a.w, a.H, a.H, a.mw, n0, a.N);
    mm_stage_out<MWMAX>(S.out, acc, a.mw);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[3] = {0.f, 0.f, 0.f};
    for (int nl = wave; nl < TN; nl += 4) {
        const int64_t n = n0 + nl;
        if (n >= a.N) break;
        for (int m = lane; m < a.H; m += 64) {
            int32_t g = sat(asr(S.out[nl][m], rs), a.out_bits);
            g = sat(wadd(g, chexp(a.bias[m], a.b_bits, a.b_exp, a.out_exp)), a.out_bits);
            if (a.tr_out2) a.tr_out2[n * a.H + m] = g;
            const int32_t s = sigmoid_lut(g, a.out_bits, a.out_exp, a.sig_x, a.sig_y, lut);
            if (a.tr_sig) a.tr_sig[n * a.H + m] = s;
            const int32_t l = chcfg(a.x1[n * a.H + m], a.y_bits, a.y_exp, a.l_bits, a.l_exp);
            const int32_t r = chcfg(s, a.out_bits, a.sig_y, a.r_bits, a.r_exp);
            const int32_t z = sat(asr(wmul(l, r), a.rs_gate), a.res_bits);
            a.z[n * a.H + m] = z;
            const float fz = tofloat(z, a.res_exp), fs = tofloat(a.skip[n * a.H + m], skip_e);
            v[0] = fmaxf(v[0], fabsf(__fadd_rn(fz, fs)));
            v[1] = fmaxf(v[1], fabsf(fz));
            v[2] = fmaxf(v[2], fabsf(fs));
        }
    }
    block_max_atomic<3>(v, a.dynw->mx + 8);
}

// residual add (compute_best) + ReLU, fxpmodel.py:1147-1159
__global__ __launch_bounds__(256) void k_resid(const int32_t *__restrict__ z, const int32_t *__restrict__ skip,
                                               int32_t *__restrict__ out, int32_t *tr_resid, int64_t NH, int res_bits,
                                               int skip_bits, const LayerDyn *dyn)
{
    const AddCb p = dyn->res;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < NH; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t r = add_cb_apply(z[i], res_bits, skip[i], skip_bits, p, res_bits);
        if (tr_resid) tr_resid[i] = r;
        out[i] = r < 0 ? 0 : r;
    }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone element-wise ops for the FxpArray-level API
// ---------------------------------------------------------------------------------------------
#define S5_GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void k_from_fp(const float *__restrict__ x, int32_t *__restrict__ y, int64_t n, int bits, int exp, int mode)
{
    const float sc = ldexpf(1.f, exp);
    S5_GRID_STRIDE(i, n)
    {
        const float v = __fmul_rn(x[i], sc);
        const float r = mode == 2 ? rintf(v) : (mode == 1 ? ceilf(v) : floorf(v));
        y[i] = sat(f2i(r), bits);
    }
}

__global__ void k_to_float(const int32_t *__restrict__ x, float *__restrict__ y, int64_t n, int exp)
{
    S5_GRID_STRIDE(i, n) y[i] = tofloat(x[i], exp);
}

__global__ void k_change_cfg(const int32_t *__restrict__ x, int32_t *__restrict__ y, int64_t n, int bits, int exp,
                             int bits2, int exp2)
{
    S5_GRID_STRIDE(i, n) y[i] = chcfg(x[i], bits, exp, bits2, exp2);
}

__global__ void k_add(const int32_t *__restrict__ x, const int32_t *__restrict__ y, int32_t *__restrict__ out,
                      int64_t n, int64_t ylen, int xb, int xe, int yb, int ye, int ob, int oe, int negy)
{
    S5_GRID_STRIDE(i, n)
    {
        int32_t yv = y[ylen == n ? i : i % ylen];
        if (negy) yv = wmul(yv, -1);
        out[i] = sat(wadd(chexp(x[i], xb, xe, oe), chexp(yv, yb, ye, oe)), ob);
    }
}

__global__ void k_mul(const int32_t *__restrict__ x, const int32_t *__restrict__ y, int32_t *__restrict__ out,
                      int64_t n, int64_t ylen, int rs, int ob)
{
    S5_GRID_STRIDE(i, n) out[i] = sat(asr(wmul(x[i], y[ylen == n ? i : i % ylen]), rs), ob);
}

// scratch: uint32[4] = {max|fx (+|*) fy|, max|fx|, max|fy|, unused}
template <bool MUL>
__global__ __launch_bounds__(256) void k_cb_reduce(const int32_t *__restrict__ x, const int32_t *__restrict__ y,
                                                   int64_t n, int64_t ylen, int xe, int ye, uint32_t *scratch)
{
    float v[3] = {0.f, 0.f, 0.f};
    S5_GRID_STRIDE(i, n)
    {
        const float fx = tofloat(x[i], xe), fy = tofloat(y[ylen == n ? i : i % ylen], ye);
        v[0] = fmaxf(v[0], fabsf(MUL ? __fmul_rn(fx, fy) : __fadd_rn(fx, fy)));
        v[1] = fmaxf(v[1], fabsf(fx));
        v[2] = fmaxf(v[2], fabsf(fy));
    }
    block_max_atomic<3>(v, scratch);
}

// out_exp_dev: {result_exp, status bits}; params (AddCb or {rs}) are left in scratch[4..7]
__global__ void k_cb_finalize(uint32_t *scratch, int xe, int ye, int ob, int is_mul, int32_t *out_exp_dev)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t st = 0;
    if (is_mul) {
        int32_t rs, eo;
        finalize_mul_cb(scratch[0], xe, ye, ob, rs, eo, &st);
        scratch[4] = (uint32_t)rs;
        out_exp_dev[0] = eo;
    } else {
        const AddCb p = finalize_add_cb(scratch, xe, ye, ob, &st);
        scratch[4] = (uint32_t)p.shx;
        scratch[5] = (uint32_t)p.shy;
        scratch[6] = (uint32_t)p.post;
        scratch[7] = (uint32_t)p.eo;
        out_exp_dev[0] = p.eo;
    }
    out_exp_dev[1] = st;
}

template <bool MUL>
__global__ void k_cb_apply(const int32_t *__restrict__ x, const int32_t *__restrict__ y, int32_t *__restrict__ out,
                           int64_t n, int64_t ylen, int xb, int yb, int ob, const uint32_t *scratch)
{
    AddCb p;
    p.shx = (int32_t)scratch[4];
    p.shy = (int32_t)scratch[5];
    p.post = (int32_t)scratch[6];
    p.eo = (int32_t)scratch[7];
    const int rs = (int32_t)scratch[4];
    S5_GRID_STRIDE(i, n)
    {
        const int32_t yv = y[ylen == n ? i : i % ylen];
        out[i] = MUL ? sat(asr(wmul(x[i], yv), rs), ob) : add_cb_apply(x[i], xb, yv, yb, p, ob);
    }
}

__global__ void k_relu(const int32_t *__restrict__ re, const int32_t *__restrict__ im, int32_t *__restrict__ ore,
                       int32_t *__restrict__ oim, int64_t n)
{
    S5_GRID_STRIDE(i, n)
    {
        if (im) {
            int32_t r = re[i], q = im[i];
            crelu(r, q);
            ore[i] = r;
            oim[i] = q;
        } else {
            const int32_t r = re[i];
            ore[i] = r < 0 ? 0 : r;
        }
    }
}

struct Lut8 {
    int32_t v[8];
};
__global__ void k_sigmoid(const int32_t *__restrict__ x, int32_t *__restrict__ y, int64_t n, int xb, int xe, int sx,
                          int sy, Lut8 lut)
{
    __shared__ int32_t l[8];
    if (threadIdx.x < 8) l[threadIdx.x] = lut.v[threadIdx.x];
    __syncthreads();
    S5_GRID_STRIDE(i, n) y[i] = sigmoid_lut(x[i], xb, xe, sx, sy, l);
}

} // namespace s5
