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
// Orientation: D^T = W^T (A operand, rows = output channels) x X^T (B operand, columns = frames).
// A lane then owns ONE frame and, per 32-channel tile, four groups of four consecutive channels
// (rows (i&3) + 8*(i>>2) + 4*(lane>>5)), so every fused epilogue runs per frame with 8-byte int16
// stores.  Activation fragments come straight from global memory (16 consecutive k per lane);
// the small weight matrix and the per-channel constants are staged once per workgroup in LDS
// (weights as [channel][k] with a row stride whose 16-byte count is odd: conflict-free ds_read_b128).
//
// Activations between these kernels are int16 (N,H); the recurrence streams stay int32 (scan_quad.hpp).
// Register discipline: fragment loads run as explicit 2-deep pipelines and the unrolled epilogues are
// fenced with sched_barrier, otherwise the compiler hoists every load of the unrolled body and spills.
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

#define S5_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ unsigned perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }

// 16 values held in int32 registers -> byte planes (4 packed registers each)
__device__ __forceinline__ void planes_from_i32(const int32_t (&v)[16], v4i &hi, v4i &lo)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned p01 = perm((unsigned)v[4 * j + 1], (unsigned)v[4 * j], 0x05010400u);     // a0l a1l a0h a1h
        const unsigned p23 = perm((unsigned)v[4 * j + 3], (unsigned)v[4 * j + 2], 0x05010400u); // a2l a3l a2h a3h
        lo[j] = (int)(perm(p23, p01, 0x05040100u) ^ 0x80808080u);
        hi[j] = (int)perm(p23, p01, 0x07060302u);
    }
}

// 16 int16 values packed two per register (as they lie in memory) -> byte planes
__device__ __forceinline__ void planes_from_i16(const v4i &w0, const v4i &w1, v4i &hi, v4i &lo)
{
    const unsigned r[8] = {(unsigned)w0[0], (unsigned)w0[1], (unsigned)w0[2], (unsigned)w0[3],
                           (unsigned)w1[0], (unsigned)w1[1], (unsigned)w1[2], (unsigned)w1[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lo[j] = (int)(perm(r[2 * j + 1], r[2 * j], 0x06040200u) ^ 0x80808080u);
        hi[j] = (int)perm(r[2 * j + 1], r[2 * j], 0x07050301u);
    }
}

__device__ __forceinline__ void unpack_i16(const v4i &w0, const v4i &w1, int32_t (&v)[16])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = (int32_t)(int16_t)(w0[j] & 0xffff);
        v[2 * j + 1] = w0[j] >> 16;
        v[8 + 2 * j] = (int32_t)(int16_t)(w1[j] & 0xffff);
        v[8 + 2 * j + 1] = w1[j] >> 16;
    }
}

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

// cooperative global -> LDS copy (bytes % 16 == 0, both 16-byte aligned)
__device__ __forceinline__ void stage_lds(void *dst, const void *src, int bytes)
{
    for (int o = threadIdx.x * 16; o < bytes; o += blockDim.x * 16)
        *reinterpret_cast<v4i *>(reinterpret_cast<int8_t *>(dst) + o) =
            *reinterpret_cast<const v4i *>(reinterpret_cast<const int8_t *>(src) + o);
}

// acc[ct] = sum_k W[ch][k] * a[k][frame] + cs[ch] for NT column tiles starting at tile ct0.
// cs (LDS, Np ints) is the per-channel constant 128*colsum; it rides on the shift between the passes.
template <int KS, int NT>
__device__ __forceinline__ void mfma_2plane(v16i (&acc)[NT], const int8_t *Wl, int Kp, const int32_t *cs, int ct0,
                                            const v4i (&hi)[KS], const v4i (&lo)[KS])
{
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5;
    const int8_t *wrow = Wl + (size_t)(32 * ct0 + r) * Kp + 16 * h;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const v4i w = *reinterpret_cast<const v4i *>(wrow + (size_t)32 * ct * Kp + 32 * ks);
            acc[ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, hi[ks], acc[ct], 0, 0, 0);
        }
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4i c = *reinterpret_cast<const v4i *>(cs + 32 * (ct0 + ct) + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ct][4 * g + e] = wadd(wshl(acc[ct][4 * g + e], 8), c[e]);
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const v4i w = *reinterpret_cast<const v4i *>(wrow + (size_t)32 * ct * Kp + 32 * ks);
            acc[ct] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, lo[ks], acc[ct], 0, 0, 0);
        }
}

// first of the four consecutive channels that accumulator elements 4g..4g+3 of column tile ct hold
__device__ __forceinline__ int acc_channel(int ct, int g) { return 32 * ct + 8 * g + 4 * ((threadIdx.x & 63) >> 5); }

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

template <int KS, int NT>
// one wave per SIMD: the 4-deep load ring (256 B per lane in flight) hides HBM latency by itself, and the
// 9 k-steps of byte planes + accumulators do not fit 256 registers without spilling
__global__ __launch_bounds__(256, 1) void k_enc_mfma(EncArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    constexpr int DEPTH = 4; // k-steps (64 bytes per lane each) in flight
    const int wbytes = a.w.Np * a.w.Kp;
    int32_t *cs = reinterpret_cast<int32_t *>(smem + wbytes), *be = cs + a.w.Np;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    const uint64_t total_bytes = (uint64_t)a.N * a.K * 4;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.x), 0,
                                                  (int)(total_bytes > 0xfffffff0ull ? 0xfffffff0ull : total_bytes), 0x00020000);
    // beyond the row end a load reads the next frame (multiplied by zero weights); beyond the tensor, 0
    v4i buf[DEPTH][4];
    auto issue = [&](unsigned row_off, int ks, v4i (&b)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            b[q] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rsrc, row_off + (unsigned)(128 * ks + 16 * q), 0, 0));
    };
    auto row_of = [&](int64_t tl) {
        const int64_t n = tl * 32 + r;
        return (unsigned)((n < a.N ? n : a.N - 1) * a.K * 4) + (unsigned)(16 * h * 4);
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < tiles) {
        const unsigned ro = row_of(tile);
#pragma unroll
        for (int ks = 0; ks < DEPTH && ks < KS; ++ks) issue(ro, ks, buf[ks]);
    }
    stage_lds(smem, a.w.wt, wbytes);
    stage_lds(cs, a.w.cs128, a.w.Np * 4);
    stage_lds(be, a.bias_eff, a.w.Np * 4);
    __syncthreads();
    bool wide = false;
    for (; tile < tiles; tile += stride) {
        const int64_t n = tile * 32 + r;
        const unsigned ro = row_of(tile);
        const bool more = tile + stride < tiles;
        const unsigned ro_next = more ? row_of(tile + stride) : ro;
        v4i hi[KS], lo[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            int32_t v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4i t = buf[ks % DEPTH][q];
                v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
            }
            // refill the slot just consumed: the rest of this tile first, then the head of the next one
            if (ks + DEPTH < KS) issue(ro, ks + DEPTH, buf[ks % DEPTH]);
            else if (more) issue(ro_next, ks + DEPTH - KS, buf[ks % DEPTH]);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (a.conv) v[j] = chcfg(v[j], a.xb, a.xe, a.inp_bits, a.inp_exp);
                wide |= (v[j] != (int32_t)(int16_t)v[j]);
            }
            planes_from_i32(v, hi[ks], lo[ks]);
            S5_FENCE();
        }
        v16i acc[NT];
        mfma_2plane<KS, NT>(acc, smem, a.w.Kp, cs, 0, hi, lo);
        S5_FENCE();
        if (n < a.N) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = acc_channel(ct, g);
                    if (ch < a.M) {
                        const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                        int32_t o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int32_t v = sat(asr(acc[ct][4 * g + e], a.rs), a.out_bits);
                            v = sat(wadd(v, bv[e]), a.out_bits);
                            o[e] = v < 0 ? 0 : v;
                        }
                        *reinterpret_cast<v2i *>(a.y + n * a.M + ch) = pack4_i16(o[0], o[1], o[2], o[3]);
                    }
                }
                S5_FENCE();
            }
        }
    }
    if (__any(wide) && l == 0) atomicOr(a.status, ST_WIDE_INPUT);
}

// ---------------------------------------------------------------------------------------------
// B projection: BN chain + change_cfg -> u, Bu = u @ [B_re^T | B_im^T], written to the scan-native
// stream already shifted to the state exponent.  fxpmodel.py:620-644, 158-167.
// LDS: [weights][cs128 Np]
// ---------------------------------------------------------------------------------------------
struct BprojMArgs {
    BnArgs bn;
    const int16_t *x; // (N,H)
    MfmaW w;          // 2P channels: [0,P) = B_re rows, [P,2P) = B_im rows
    int32_t *bq;      // native stream
    int32_t *tr_bu_re, *tr_bu_im, *tr_pre_s5, *tr_u; // optional int32 traces
    int64_t N;
    int32_t L, TB, H, P;
    int32_t rs_re, rs_im, bre_bits, bim_bits, sh_re, sh_im;
};

template <int KS, int NT>
__global__ __launch_bounds__(256, 2) void k_bproj_mfma(BprojMArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int wbytes = a.w.Np * a.w.Kp;
    int32_t *cs = reinterpret_cast<int32_t *>(smem + wbytes);
    stage_lds(smem, a.w.wt, wbytes);
    stage_lds(cs, a.w.cs128, a.w.Np * 4);
    __syncthreads();
    const LayerDyn d = *a.bn.dyn;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n = tile * 32 + r;
        const int64_t nn = n < a.N ? n : a.N - 1;
        v4i hi[KS], lo[KS];
        v4i raw[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 16 * h;
            raw[ks][0] = *reinterpret_cast<const v4i *>(a.x + nn * a.H + k0);
            raw[ks][1] = *reinterpret_cast<const v4i *>(a.x + nn * a.H + k0 + 8);
        }
        S5_FENCE();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 16 * h;
            int32_t v[16];
            unpack_i16(raw[ks][0], raw[ks][1], v);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int32_t t = bn_chain<4>(a.bn, d, v[j], k0 + j);
                const int32_t u = chcfg(t, a.bn.out_bits, d.bn_e, a.bn.ub, a.bn.ue);
                if (a.tr_pre_s5 && n < a.N) a.tr_pre_s5[n * a.H + k0 + j] = t;
                if (a.tr_u && n < a.N) a.tr_u[n * a.H + k0 + j] = u;
                v[j] = u;
            }
            planes_from_i32(v, hi[ks], lo[ks]);
            S5_FENCE();
        }
        v16i acc[NT];
        mfma_2plane<KS, NT>(acc, smem, a.w.Kp, cs, 0, hi, lo);
        S5_FENCE();
        if (n < a.N) {
            const int64_t b = n / a.L;
            const int t = (int)(n - b * a.L);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = acc_channel(ct, g);
                    if (ch < 2 * a.P) {
                        const int c = ch >= a.P;
                        const int p = ch - c * a.P;
                        const int rs = c ? a.rs_im : a.rs_re, bits = c ? a.bim_bits : a.bre_bits, sh = c ? a.sh_im : a.sh_re;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int32_t bu = sat(asr(acc[ct][4 * g + e], rs), bits);
                            a.bq[native_word(b, t, p + e, c, a.TB, a.P)] = sh > 0 ? asr(bu, sh) : wshl(bu, -sh);
                            if (!c && a.tr_bu_re) a.tr_bu_re[n * a.P + p + e] = bu;
                            if (c && a.tr_bu_im) a.tr_bu_im[n * a.P + p + e] = bu;
                        }
                    }
                }
                S5_FENCE();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// C projection + D*u + ReLU from the RAW native state stream.  fxpmodel.py:740-793, 1125.
// Each wave transposes its 32 frames through LDS: coalesced 32-byte (re,im) chunks of 4 steps in,
// complex ReLU + range check, int16 [frame][comp][state] out.
// LDS: [W_re][W_im][cs_re Np][cs_im Np][D Np][4 wave tiles]
// ---------------------------------------------------------------------------------------------
struct CprojMArgs {
    BnArgs bn;
    const int16_t *x;    // (N,H): the SSM input u (have_u) or the layer input from which u is recomputed
    int32_t have_u;
    const int32_t *xs;   // native raw states
    MfmaW w_re, w_im;    // H channels each, K = P
    const int32_t *D;    // (Np)
    int16_t *x1;         // (N,H) relu(ys)
    int32_t *tr_ys;      // optional (N,H)
    int64_t N;
    int32_t L, TB, H, P;
    int32_t rs_re, rs_im, rs_d, y_bits;
    int32_t xmax;
    LayerDyn *dynw;
    int32_t *status;
};

template <int KS, int NT>
__global__ __launch_bounds__(256, 2) void k_cproj_mfma(CprojMArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    constexpr int P = 32 * KS;         // K of this projection is the state count
    constexpr int ITER = 8 * P / 64;   // 32-byte (state, 4 steps, re+im) chunks per lane and tile
    constexpr int BATCH = 8;           // chunks in flight per lane
    const int wbytes = a.w_re.Np * a.w_re.Kp, Np = a.w_re.Np;
    int8_t *Wre = smem, *Wim = smem + wbytes;
    int32_t *csr = reinterpret_cast<int32_t *>(smem + 2 * wbytes), *csi = csr + Np, *Dl = csi + Np;
    constexpr int row_bytes = 2 * P * 2 + 16; // [comp][state] int16 + pad (odd number of 16-byte slots)
    int8_t *tile = reinterpret_cast<int8_t *>(Dl + Np) + (threadIdx.x >> 6) * 32 * row_bytes;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile_i = (int64_t)blockIdx.x * 4 + wave;
    // epilogue operand (u or the layer input), requested with the tile's first loads
    v2i uq[NT][4];
    auto fetch_u = [&](int64_t tl) {
        int64_t n = tl * 32 + r;
        n = n < a.N ? n : a.N - 1;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) uq[ct][g] = *reinterpret_cast<const v2i *>(a.x + n * a.H + 32 * ct + 8 * g + 4 * h);
    };
    if (tile_i < tiles) fetch_u(tile_i);
    stage_lds(Wre, a.w_re.wt, wbytes);
    stage_lds(Wim, a.w_im.wt, wbytes);
    stage_lds(csr, a.w_re.cs128, Np * 4);
    stage_lds(csi, a.w_im.cs128, Np * 4);
    stage_lds(Dl, a.D, Np * 4);
    __syncthreads();
    const LayerDyn d = *a.bn.dyn;
    bool bad = false;
    for (; tile_i < tiles; tile_i += stride) {
        const int64_t n0 = tile_i * 32;
        // ---- stage: 8 groups of 4 frames x P states, coalesced 32-byte chunks, BATCH of them in flight per lane
#pragma unroll
        for (int bt = 0; bt < ITER / BATCH; ++bt) {
            v4i cre[BATCH], cim[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int q = l + 64 * (bt * BATCH + i);
                const int grp = q / P, p = q % P;
                int64_t nf = n0 + 4 * grp; // first frame of the group (L % 4 == 0: a group never straddles sequences)
                nf = nf < a.N ? nf : a.N - 4;
                const int64_t b = nf / a.L;
                const int t = (int)(nf - b * a.L);
                const int32_t *src = a.xs + native_word(b, t, p, 0, a.TB, P);
                cre[i] = *reinterpret_cast<const v4i *>(src);
                cim[i] = *reinterpret_cast<const v4i *>(src + 4);
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int q = l + 64 * (bt * BATCH + i);
                const int grp = q / P, p = q % P;
                if (n0 + 4 * grp < a.N) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int32_t xr = cre[i][j], xi = cim[i][j];
                        bad |= (xr > a.xmax) | (xr < -a.xmax) | (xi > a.xmax) | (xi < -a.xmax);
                        crelu(xr, xi);
                        int8_t *row = tile + (4 * grp + j) * row_bytes;
                        *reinterpret_cast<int16_t *>(row + 2 * p) = (int16_t)xr;
                        *reinterpret_cast<int16_t *>(row + 2 * (P + p)) = (int16_t)xi;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's LDS writes are done before its own reads
        const int64_t n = n0 + r;
        v4i hr[KS], lr[KS], hm[KS], lm[KS];
        const int8_t *row = tile + r * row_bytes;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 16 * h;
            planes_from_i16(*reinterpret_cast<const v4i *>(row + 2 * k0), *reinterpret_cast<const v4i *>(row + 2 * k0 + 16), hr[ks], lr[ks]);
            planes_from_i16(*reinterpret_cast<const v4i *>(row + 2 * (P + k0)), *reinterpret_cast<const v4i *>(row + 2 * (P + k0) + 16), hm[ks], lm[ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... and its reads before the next tile's writes
        v16i are[NT], aim[NT];
        mfma_2plane<KS, NT>(are, Wre, a.w_re.Kp, csr, 0, hr, lr);
        mfma_2plane<KS, NT>(aim, Wim, a.w_im.Kp, csi, 0, hm, lm);
        S5_FENCE();
        if (n < a.N) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = acc_channel(ct, g);
                    if (ch < a.H) {
                        const v4i Dv = *reinterpret_cast<const v4i *>(Dl + ch);
                        int32_t hv[4], o[4];
                        unpack4_i16(uq[ct][g], hv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int32_t cr = sat(asr(are[ct][4 * g + e], a.rs_re), a.y_bits);
                            const int32_t ci = sat(asr(aim[ct][4 * g + e], a.rs_im), a.y_bits);
                            const int32_t cx = sat(wadd(cr, wmul(ci, -1)), a.y_bits);
                            const int32_t cx2 = wmul(cx, 2); // not clipped, fxpmodel.py:765-767
                            // x points at the stored SSM input u when have_u, else at the layer input (chain recomputed)
                            const int32_t u = a.have_u ? hv[e] : bn_chain<5>(a.bn, d, hv[e], ch + e);
                            const int32_t du = sat(asr(wmul(Dv[e], u), a.rs_d), a.y_bits);
                            const int32_t y = sat(wadd(cx2, du), a.y_bits);
                            if (a.tr_ys) a.tr_ys[n * a.H + ch + e] = y;
                            o[e] = y < 0 ? 0 : y;
                        }
                        *reinterpret_cast<v2i *>(a.x1 + n * a.H + ch) = pack4_i16(o[0], o[1], o[2], o[3]);
                    }
                }
                S5_FENCE();
            }
        }
        if (tile_i + stride < tiles) fetch_u(tile_i + stride);
        __builtin_amdgcn_wave_barrier();
    }
    if (__any(bad) && l == 0) {
        atomicExch(&a.dynw->redo, 1);
        atomicOr(a.status, ST_WIDE_STATE);
    }
}

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

template <int KS, int NT>
__global__ __launch_bounds__(256, 2) void k_out2gate_mfma(GateMArgs a)
{
    if (a.run_if && *a.run_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int wbytes = a.w.Np * a.w.Kp;
    int32_t *cs = reinterpret_cast<int32_t *>(smem + wbytes), *be = cs + a.w.Np, *lut = be + a.w.Np;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    // Every global load of a tile -- MFMA fragments and the epilogue's x1 / skip operands -- is issued in one
    // go (for the first tile even before the weights are staged), so a tile costs one round of memory latency.
    v4i raw[KS][2];
    v2i xq[NT][4], sq[NT][4];
    auto fetch = [&](int64_t tl) {
        int64_t n = tl * 32 + r;
        n = n < a.N ? n : a.N - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            raw[ks][0] = *reinterpret_cast<const v4i *>(a.x1 + n * a.H + 32 * ks + 16 * h);
            raw[ks][1] = *reinterpret_cast<const v4i *>(a.x1 + n * a.H + 32 * ks + 16 * h + 8);
        }
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xq[ct][g] = *reinterpret_cast<const v2i *>(a.x1 + n * a.H + 32 * ct + 8 * g + 4 * h);
                sq[ct][g] = *reinterpret_cast<const v2i *>(a.skip + n * a.H + 32 * ct + 8 * g + 4 * h);
            }
    };
    if (tile < tiles) fetch(tile);
    if (threadIdx.x < 8) lut[threadIdx.x] = a.lut[threadIdx.x];
    stage_lds(smem, a.w.wt, wbytes);
    stage_lds(cs, a.w.cs128, a.w.Np * 4);
    stage_lds(be, a.bias_eff, a.w.Np * 4);
    __syncthreads();
    const int skip_e = a.skip_e.get();
    float mx[3] = {0.f, 0.f, 0.f};
    for (; tile < tiles; tile += stride) {
        const int64_t n = tile * 32 + r;
        v4i hi[KS], lo[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (a.conv) {
                int32_t v[16];
                unpack_i16(raw[ks][0], raw[ks][1], v);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = chcfg(v[j], a.y_bits, a.y_exp, a.inp_bits, a.inp_exp);
                planes_from_i32(v, hi[ks], lo[ks]);
            } else {
                planes_from_i16(raw[ks][0], raw[ks][1], hi[ks], lo[ks]);
            }
        }
        S5_FENCE();
        v16i acc[NT];
        mfma_2plane<KS, NT>(acc, smem, a.w.Kp, cs, 0, hi, lo);
        S5_FENCE();
        if (n < a.N) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = acc_channel(ct, g);
                    if (ch < a.H) {
                        const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                        int32_t xv[4], sv[4], o[4];
                        unpack4_i16(xq[ct][g], xv);
                        unpack4_i16(sq[ct][g], sv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int32_t gq = sat(asr(acc[ct][4 * g + e], a.rs), a.out_bits);
                            gq = sat(wadd(gq, bv[e]), a.out_bits);
                            if (a.tr_out2) a.tr_out2[n * a.H + ch + e] = gq;
                            const int32_t s = sigmoid_lut(gq, a.out_bits, a.out_exp, a.sig_x, a.sig_y, lut);
                            if (a.tr_sig) a.tr_sig[n * a.H + ch + e] = s;
                            const int32_t lq = chcfg(xv[e], a.y_bits, a.y_exp, a.l_bits, a.l_exp);
                            const int32_t rq = chcfg(s, a.out_bits, a.sig_y, a.r_bits, a.r_exp);
                            const int32_t z = sat(asr(wmul(lq, rq), a.rs_gate), a.res_bits);
                            if (a.tr_z) a.tr_z[n * a.H + ch + e] = z;
                            o[e] = z;
                            const float fz = tofloat(z, a.res_exp), fs = tofloat(sv[e], skip_e);
                            mx[0] = fmaxf(mx[0], fabsf(__fadd_rn(fz, fs)));
                            mx[1] = fmaxf(mx[1], fabsf(fz));
                            mx[2] = fmaxf(mx[2], fabsf(fs));
                        }
                        *reinterpret_cast<v2i *>(a.z + n * a.H + ch) = pack4_i16(o[0], o[1], o[2], o[3]);
                    }
                }
                S5_FENCE();
            }
        }
        if (tile + stride < tiles) fetch(tile + stride);
    }
    block_max_atomic<3>(mx, a.dynw->mx + a.mx_slot);
}

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

template <int KS, int NT, int CG>
__global__ __launch_bounds__(256, 2) void k_dec_mfma(DecArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int wbytes = a.w.Np * a.w.Kp;
    int32_t *cs = reinterpret_cast<int32_t *>(smem + wbytes), *be = cs + a.w.Np;
    stage_lds(smem, a.w.wt, wbytes);
    stage_lds(cs, a.w.cs128, a.w.Np * 4);
    stage_lds(be, a.bias_eff, a.w.Np * 4);
    __syncthreads();
    const int xe0 = a.xe.get();
    const bool conv = a.xb > a.inp_bits || xe0 > a.inp_exp;
    int rs = (conv ? a.inp_exp : xe0) + a.w_exp - a.out_exp;
    if (rs < 0 || rs > 31) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(a.status, ST_NEGSHIFT);
        rs = rs < 0 ? 0 : 31;
    }
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n = tile * 32 + r;
        const int64_t nn = n < a.N ? n : a.N - 1;
        v4i hi[KS], lo[KS];
        v4i raw[KS][2];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = 32 * ks + 16 * h;
            raw[ks][0] = *reinterpret_cast<const v4i *>(a.x + nn * a.H + k0);
            raw[ks][1] = *reinterpret_cast<const v4i *>(a.x + nn * a.H + k0 + 8);
        }
        S5_FENCE();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (conv) {
                int32_t v[16];
                unpack_i16(raw[ks][0], raw[ks][1], v);
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = chcfg(v[j], a.xb, xe0, a.inp_bits, a.inp_exp);
                planes_from_i32(v, hi[ks], lo[ks]);
            } else {
                planes_from_i16(raw[ks][0], raw[ks][1], hi[ks], lo[ks]);
            }
        }
        S5_FENCE();
#pragma unroll 1
        for (int cg = 0; cg < CG; ++cg) {
            v16i acc[NT];
            mfma_2plane<KS, NT>(acc, smem, a.w.Kp, cs, cg * NT, hi, lo);
            S5_FENCE();
            if (n < a.N) {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch = acc_channel(cg * NT + ct, g);
                        if (ch < a.M) {
                            const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                            v4i o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int32_t v = sat(asr(acc[ct][4 * g + e], rs), a.out_bits);
                                o[e] = sat(wadd(v, bv[e]), a.out_bits);
                            }
                            int32_t *dst = a.y + n * a.M + ch;
                            if (ch + 4 <= a.M) *reinterpret_cast<v4i *>(dst) = o; // 4-byte aligned 16-byte store
                            else
                                for (int e = 0; e < 4 && ch + e < a.M; ++e) dst[e] = o[e];
                        }
                    }
                    S5_FENCE();
                }
            }
        }
    }
}

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
