// mfma_fused.hpp -- C projection, D*u, ReLU, out2 dense, LUT sigmoid and mult_gate in ONE kernel.
//
// fxpmodel.py:740-793 (C projection, x2, + D*u), :1125 (ReLU), :1133-1137 (out2, sigmoid, gate), plus the
// float32 maxima of the residual compute_best add (:1147-1152).
//
// The C projection's result tile lives in accumulator layout: lane = frame, registers = the channels
// 32*ct + 8*g + 4*h + e.  The out2 matmul sums over exactly those channels, and an MFMA may visit k in any
// order as long as both operands agree -- so the host stores the out2 weights with the k-order of each
// 32-block permuted to (h, g, e) (pack_mfma_kperm) and x1 = relu(y) goes from the first epilogue straight
// into the second MFMA's B operand: no LDS, no memory round trip for x1, and the gate finds x1 and the out2
// result in the same lane and register position.
//
// LDS: [W_re][W_im][W_out2 (k-permuted)][cs_re][cs_im][D][cs_out2][bias_eff][lut 8][4 wave tiles]
#pragma once
#include "proj_p.hpp"

namespace s5 {

struct CGateArgs {
    const int16_t *u;    // (N,H) SSM input (written by the B projection)
    const int16_t *skip; // (N,H) layer input
    const int32_t *xs;   // native raw states
    MfmaW w_re, w_im;    // H channels each, K = P
    MfmaW w_o2;          // H channels, K = H, k-permuted
    const int32_t *D;    // (Np)
    const int32_t *bias_eff; // (Np) out2 bias at out_exp
    int16_t *z;          // (N,H)
    int32_t *tr_ys, *tr_out2, *tr_sig, *tr_z; // traces (TRACE instantiation only)
    int64_t N;
    int32_t L, TB, H;
    int32_t rs_re, rs_im, rs_d, y_bits, y_exp;
    int32_t xmax;
    int32_t conv, inp_bits, inp_exp, rs_o2, out_bits, out_exp;
    int32_t sig_x, sig_y;
    int32_t lut[8];
    int32_t l_bits, l_exp, r_bits, r_exp, res_bits, res_exp, rs_gate;
    DynExp skip_e;
    LayerDyn *dynw;
    int32_t *status;
};

// multi-rank mode only: the residual maxima of a re-run layer live in slots 11..13; move them to 8..10, the
// slots the ranks exchange
__global__ void k_select_maxima(LayerDyn *d)
{
    if (threadIdx.x < 3 && blockIdx.x == 0 && d->redo) d->mx[8 + threadIdx.x] = d->mx[11 + threadIdx.x];
}

template <int KS, int NT, bool TRACE>
__global__ __launch_bounds__(256, 2) void k_cgate_mfma(CGateArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    constexpr int P = 32 * KS;
    constexpr int ITER = 8 * P / 64, BATCH = 8;
    constexpr int row_bytes = 2 * P * 2 + 16;
    const int wb = a.w_re.Np * a.w_re.Kp, wb2 = a.w_o2.Np * a.w_o2.Kp, Np = a.w_re.Np;
    int8_t *Wre = smem, *Wim = smem + wb, *Wo2 = smem + 2 * wb;
    int32_t *csr = reinterpret_cast<int32_t *>(smem + 2 * wb + wb2), *csi = csr + Np, *Dl = csi + Np, *cs2 = Dl + Np,
            *be = cs2 + Np, *lut = be + Np;
    int8_t *tile = reinterpret_cast<int8_t *>(lut + 8) + (threadIdx.x >> 6) * 32 * row_bytes;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int64_t tiles = (a.N + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t tile_i = (int64_t)blockIdx.x * 4 + wave;
    v2i uq[NT][4], sq[NT][4];
    auto fetch_ops = [&](int64_t tl) {
        int64_t n = tl * 32 + r;
        n = n < a.N ? n : a.N - 1;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uq[ct][g] = *reinterpret_cast<const v2i *>(a.u + n * a.H + 32 * ct + 8 * g + 4 * h);
                sq[ct][g] = *reinterpret_cast<const v2i *>(a.skip + n * a.H + 32 * ct + 8 * g + 4 * h);
            }
    };
    if (tile_i < tiles) fetch_ops(tile_i);
    if (threadIdx.x < 8) lut[threadIdx.x] = a.lut[threadIdx.x];
    stage_lds(Wre, a.w_re.wt, wb);
    stage_lds(Wim, a.w_im.wt, wb);
    stage_lds(Wo2, a.w_o2.wt, wb2);
    stage_lds(csr, a.w_re.cs128, Np * 4);
    stage_lds(csi, a.w_im.cs128, Np * 4);
    stage_lds(Dl, a.D, Np * 4);
    stage_lds(cs2, a.w_o2.cs128, Np * 4);
    stage_lds(be, a.bias_eff, Np * 4);
    __syncthreads();
    const int skip_e = a.skip_e.get();
    bool bad = false;
    float mx[3] = {0.f, 0.f, 0.f};
    for (; tile_i < tiles; tile_i += stride) {
        const int64_t n0 = tile_i * 32;
        // ---- raw states -> complex ReLU + range check -> int16 [frame][comp][state] in this wave's LDS tile
#pragma unroll
        for (int bt = 0; bt < ITER / BATCH; ++bt) {
            v4i cre[BATCH], cim[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int q = l + 64 * (bt * BATCH + i);
                const int grp = q / P, p = q % P;
                int64_t nf = n0 + 4 * grp;
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t n = n0 + r;
        int32_t x1v[NT][16];
        {
            v4i hr[KS], lr[KS], hm[KS], lm[KS];
            const int8_t *row = tile + r * row_bytes;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k0 = 32 * ks + 16 * h;
                planes_from_i16(*reinterpret_cast<const v4i *>(row + 2 * k0), *reinterpret_cast<const v4i *>(row + 2 * k0 + 16), hr[ks], lr[ks]);
                planes_from_i16(*reinterpret_cast<const v4i *>(row + 2 * (P + k0)), *reinterpret_cast<const v4i *>(row + 2 * (P + k0) + 16), hm[ks], lm[ks]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v16i are[NT], aim[NT];
            mfma_2plane<KS, NT>(are, Wre, a.w_re.Kp, csr, 0, hr, lr);
            mfma_2plane<KS, NT>(aim, Wim, a.w_im.Kp, csi, 0, hm, lm);
            S5_FENCE();
            // ---- first epilogue: y = sat(2*sat(sat(re) - sat(im)) + sat(D*u)), x1 = relu(y)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const v4i Dv = *reinterpret_cast<const v4i *>(Dl + 32 * ct + 8 * g + 4 * h);
                    int32_t uv[4];
                    unpack4_i16(uq[ct][g], uv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int32_t cr = sat(asr(are[ct][4 * g + e], a.rs_re), a.y_bits);
                        const int32_t ci = sat(asr(aim[ct][4 * g + e], a.rs_im), a.y_bits);
                        const int32_t cx = sat(wadd(cr, wmul(ci, -1)), a.y_bits);
                        const int32_t cx2 = wmul(cx, 2); // not clipped, fxpmodel.py:765-767
                        const int32_t du = sat(asr(wmul(Dv[e], uv[e]), a.rs_d), a.y_bits);
                        const int32_t y = sat(wadd(cx2, du), a.y_bits);
                        if (TRACE) {
                            if (a.tr_ys && n < a.N) a.tr_ys[n * a.H + 32 * ct + 8 * g + 4 * h + e] = y;
                        }
                        x1v[ct][4 * g + e] = y < 0 ? 0 : y;
                    }
                }
                S5_FENCE();
            }
        }
        // ---- out2: x1 (accumulator layout) is the B operand of k-step ct; weights are k-permuted to match
        v16i acc[NT];
        {
            v4i hi[NT], lo[NT]; // K = H = 32*NT
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    v[j] = a.conv ? chcfg(x1v[ct][j], a.y_bits, a.y_exp, a.inp_bits, a.inp_exp) : x1v[ct][j];
                planes_from_i32(v, hi[ct], lo[ct]);
            }
            mfma_2plane<NT, NT>(acc, Wo2, a.w_o2.Kp, cs2, 0, hi, lo);
        }
        S5_FENCE();
        // ---- second epilogue: bias, LUT sigmoid, gate with x1, maxima of z + skip
        if (n < a.N) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * ct + 8 * g + 4 * h;
                    const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                    int32_t sv[4], o[4];
                    unpack4_i16(sq[ct][g], sv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int32_t gq = sat(asr(acc[ct][4 * g + e], a.rs_o2), a.out_bits);
                        gq = sat(wadd(gq, bv[e]), a.out_bits);
                        const int32_t s = sigmoid_lut(gq, a.out_bits, a.out_exp, a.sig_x, a.sig_y, lut);
                        const int32_t lq = chcfg(x1v[ct][4 * g + e], a.y_bits, a.y_exp, a.l_bits, a.l_exp);
                        const int32_t rq = chcfg(s, a.out_bits, a.sig_y, a.r_bits, a.r_exp);
                        const int32_t z = sat(asr(wmul(lq, rq), a.rs_gate), a.res_bits);
                        if (TRACE) {
                            if (a.tr_out2) a.tr_out2[n * a.H + ch + e] = gq;
                            if (a.tr_sig) a.tr_sig[n * a.H + ch + e] = s;
                            if (a.tr_z) a.tr_z[n * a.H + ch + e] = z;
                        }
                        o[e] = z;
                        const float fz = tofloat(z, a.res_exp), fs = tofloat(sv[e], skip_e);
                        mx[0] = fmaxf(mx[0], fabsf(__fadd_rn(fz, fs)));
                        mx[1] = fmaxf(mx[1], fabsf(fz));
                        mx[2] = fmaxf(mx[2], fabsf(fs));
                    }
                    *reinterpret_cast<v2i *>(a.z + n * a.H + ch) = pack4_i16(o[0], o[1], o[2], o[3]);
                }
                S5_FENCE();
            }
        }
        if (tile_i + stride < tiles) fetch_ops(tile_i + stride);
        __builtin_amdgcn_wave_barrier();
    }
    if (__any(bad) && l == 0) {
        atomicExch(&a.dynw->redo, 1);
        atomicOr(a.status, ST_WIDE_STATE);
    }
    block_max_atomic<3>(mx, a.dynw->mx + 8);
}

} // namespace s5
