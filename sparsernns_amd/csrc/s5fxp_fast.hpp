// s5fxp_fast.hpp -- host side of the int8-MFMA forward (included by s5fxp_api.hip after Packer,
// s5fxp_model, WsLayout-independent helpers).  Packs the weights for mfma_proj.hpp and enqueues the
// w8a16 fast path:  encoder -> [BN maxima -> B proj -> recurrence -> C proj -> out2/gate -> residual] x L
// -> decoder, with int16 activations and int32 recurrence streams.
#pragma once

struct MfmaWDev {
    MfmaW w{};
    const int32_t *bias_eff = nullptr; // [Np], bias moved to out_exp (dense layers only)
};

struct FastLayer {
    MfmaWDev bproj, cre, cim, out2;
    MfmaWDev bproj_pair; // the same weights with re / im of a state 16 columns apart inside one 32-column tile
    const int32_t *Dpad = nullptr; // [Np]
    const int32_t *sigtab = nullptr; // [2][7 << sig_x] (mfma_fused.hpp k_cgate_p)
    const int16_t *sigdir = nullptr; // [1 << sigdir_bits] when the sigmoid input has <= 12 bits (DIRECT)
    int sigdir_bits = 0;
    bool bias16 = false; // every out2 bias, moved to out_exp, fits 16 bits (a condition of the gate kernel's PK16 epilogues)
    // Live-state compaction.  A state whose rows of B_bar_re and B_bar_im are all zero (8-bit B_bar with ONE exponent per matrix
    // rounds the rows of slow states away: 43-46 of 64 on the N-DNS recipe at dim_scale 0.5, profiles/r03_sparsity_census.log)
    // receives Bu = 0 at every step, so from a zero carry it stays (0, 0) for ever: asr(A * 0) = 0, the complex ReLU keeps
    // (0, 0), and its column of C multiplies zeros (fxpmodel.py:147-172, :740-763).  When at most half of a layer's states
    // are live, the layer is ALSO packed over c_slots state slots -- the live states in their order, then zero slots -- and
    // forwards that neither trace the states nor carry them in or out run every kernel of the layer on that half: half the
    // bytes of both recurrence streams, half the recurrence waves, half the gate kernel's phase A.  Bit-identical by the
    // argument above (the dropped terms are exact zeros of int32 sums).
    bool compact_ok = false;
    int n_live = 0;
    int c_slots = 0; // state slots of the compacted layer: n_live rounded up to a multiple of 32 (when that is <= P / 2)
    MfmaWDev c_bproj, c_bproj_pair, c_cre, c_cim;
    ScanBounds c_bounds; // the recurrence kernels' exactness bounds over the live states only
    const int32_t *c_a_re = nullptr, *c_a_im = nullptr; // (c_slots) Lambda_bar of the slots (0 for the empty ones)
};

struct FastModel {
    MfmaWDev enc, dec;
    std::vector<FastLayer> layers;
};

namespace {

inline const void *put_raw(Packer &p, const void *src, size_t bytes)
{
    const void *d = p.dev + p.off;
    if (p.host) std::memcpy(p.host + p.off, src, bytes);
    p.off = (p.off + bytes + 255) & ~(size_t)255;
    return d;
}

inline bool fits_bits(const int32_t *v, size_t n, int bits)
{
    const int32_t hi = (1 << (bits - 1)) - 1, lo = -hi - 1;
    for (size_t i = 0; i < n; ++i)
        if (v[i] < lo || v[i] > hi) return false;
    return true;
}

// The MFMA path is instantiated for the NDNS shapes (recipes/ndns.json at dim_scale 0.5 and 1.0) with
// <= 8-bit weights and <= 16-bit activations; everything else runs the generic kernels.
bool fast_eligible(const s5fxp_model_desc *d)
{
    if (d->n_layers < 1) return false;
    const int H = d->encoder.M, K = d->encoder.K, M = d->decoder.M, P = d->layers[0].ssm.P;
    if (!((H == 96 && P == 64) || (H == 192 && P == 128))) return false;
    if (K <= 256 || K > 288 || M > 288 || M < 1) return false;
    auto dense16 = [](const s5fxp_dense_desc &e) { return e.inp_bits <= 16 && e.out_bits <= 16 && e.b_bits <= 32; };
    if (!dense16(d->encoder) || d->decoder.inp_bits > 16 || d->decoder.out_bits > 32) return false;
    if (!fits_bits(d->encoder.weight, (size_t)K * H, 8) || !fits_bits(d->decoder.weight, (size_t)H * M, 8)) return false;
    for (int i = 0; i < d->n_layers; ++i) {
        const s5fxp_layer_desc &l = d->layers[i];
        const s5fxp_ssm_desc &s = l.ssm;
        if (s.P != P) return false;
        const s5fxp_norm_desc &n = l.norm;
        if (n.mean_bits > 16 || n.invsq_var_bits > 16 || (n.scale && n.scale_bits > 16) || (n.bias && n.bias_bits > 16))
            return false;
        if (s.u_bits > 16 || s.y_bits > 16 || s.Bu_re_bits > 32 || s.Bu_im_bits > 32) return false;
        if (!dense16(l.out2) || l.l_bits > 16 || l.r_bits > 16 || l.res_bits > 16) return false;
        if (!fits_bits(s.B_re, (size_t)P * H, 8) || !fits_bits(s.B_im, (size_t)P * H, 8) ||
            !fits_bits(s.C_re, (size_t)H * P, 8) || !fits_bits(s.C_im, (size_t)H * P, 8) ||
            !fits_bits(l.out2.weight, (size_t)H * H, 8) || !fits_bits(s.D, (size_t)H, 16))
            return false;
        for (int i8 = 0; i8 < 8; ++i8) // the fused gate kernel keeps two LUT entries per 32-bit word
            if (l.lut[i8] < 0 || l.lut[i8] > 65535) return false;
        if (l.sig_x_exp > 6) return false; // the reference's rule is min(out2.out_exp, 6) (fxpmodel.py:1097-1104); the r table assumes it
    }
    return true;
}

// get(k, ch) -> weight; result rows are channels, padded and strided for conflict-free 16-byte LDS reads
template <class Get>
void pack_mfma(Packer &p, Get get, int K, int M, MfmaWDev &o)
{
    const int Kpad = (K + 31) / 32 * 32;
    const int Kp = ((Kpad / 16) % 2 == 0) ? Kpad + 16 : Kpad;
    const int Np = (M + 31) / 32 * 32;
    std::vector<int8_t> wt((size_t)Np * Kp, 0);
    std::vector<int32_t> cs(Np, 0);
    for (int ch = 0; ch < M; ++ch) {
        uint32_t sum = 0;
        for (int k = 0; k < K; ++k) {
            const int32_t v = get(k, ch);
            wt[(size_t)ch * Kp + k] = (int8_t)v;
            sum += (uint32_t)v;
        }
        cs[ch] = (int32_t)(sum * 128u);
    }
    o.w.wt = reinterpret_cast<const int8_t *>(put_raw(p, wt.data(), wt.size()));
    o.w.cs128 = reinterpret_cast<const int32_t *>(put_raw(p, cs.data(), cs.size() * 4));
    o.w.Kp = Kp;
    o.w.Np = Np;
}

void pack_bias_eff(Packer &p, const s5fxp_dense_desc &e, int Np, MfmaWDev &o)
{
    std::vector<int32_t> b(Np, 0);
    for (int ch = 0; ch < e.M; ++ch) b[ch] = fxp::chexp(e.bias[ch], e.b_bits, e.b_exp, e.out_exp);
    o.bias_eff = reinterpret_cast<const int32_t *>(put_raw(p, b.data(), b.size() * 4));
}

void pack_fast(Packer &p, const s5fxp_model_desc *d, FastModel *f)
{
    FastModel tmp;
    if (!f) f = &tmp;
    f->layers.resize(d->n_layers);
    const s5fxp_dense_desc &e = d->encoder;
    pack_mfma(p, [&](int k, int ch) { return e.weight[(size_t)k * e.M + ch]; }, e.K, e.M, f->enc);
    pack_bias_eff(p, e, f->enc.w.Np, f->enc);
    for (int i = 0; i < d->n_layers; ++i) {
        const s5fxp_layer_desc &l = d->layers[i];
        const s5fxp_ssm_desc &s = l.ssm;
        FastLayer &o = f->layers[i];
        const int H = s.H, P = s.P;
        pack_mfma(p, [&](int k, int ch) { return ch < P ? s.B_re[(size_t)ch * H + k] : s.B_im[(size_t)(ch - P) * H + k]; }, H,
                  2 * P, o.bproj);
        pack_mfma(p, [&](int k, int ch) {
                      const int st = 16 * (ch / 32) + (ch & 15);
                      return (ch & 16) ? s.B_im[(size_t)st * H + k] : s.B_re[(size_t)st * H + k];
                  }, H, 2 * P, o.bproj_pair);
        pack_mfma(p, [&](int k, int ch) { return s.C_re[(size_t)ch * P + k]; }, P, H, o.cre);
        pack_mfma(p, [&](int k, int ch) { return s.C_im[(size_t)ch * P + k]; }, P, H, o.cim);
        pack_mfma(p, [&](int k, int ch) { return l.out2.weight[(size_t)k * l.out2.M + ch]; }, H, H, o.out2);
        pack_bias_eff(p, l.out2, o.out2.w.Np, o.out2);
        o.bias16 = true;
        for (int ch = 0; ch < l.out2.M; ++ch) {
            const int32_t v = fxp::chexp(l.out2.bias[ch], l.out2.b_bits, l.out2.b_exp, l.out2.out_exp);
            o.bias16 = o.bias16 && v >= -32768 && v <= 32767;
        }
        {
            std::vector<int> idx;
            for (int q = 0; q < P; ++q) {
                bool live = false;
                for (int k = 0; k < H && !live; ++k) live = s.B_re[(size_t)q * H + k] != 0 || s.B_im[(size_t)q * H + k] != 0;
                if (live) idx.push_back(q);
            }
            o.n_live = (int)idx.size();
            // the fewest 32-state groups that hold the live states, if that is at most half of the layer's
            const int Pc = o.n_live <= 32 ? 32 : (o.n_live + 31) / 32 * 32;
            o.compact_ok = P % 64 == 0 && Pc <= P / 2;
            o.c_slots = o.compact_ok ? Pc : P;
            if (o.compact_ok) {
                o.c_bounds = scan_bounds(s, idx.data(), o.n_live, Pc, true);
                idx.resize(Pc, -1);
                auto bre = [&](int st, int k) { return idx[st] < 0 ? 0 : s.B_re[(size_t)idx[st] * H + k]; };
                auto bim = [&](int st, int k) { return idx[st] < 0 ? 0 : s.B_im[(size_t)idx[st] * H + k]; };
                pack_mfma(p, [&](int k, int ch) { return ch < Pc ? bre(ch, k) : bim(ch - Pc, k); }, H, 2 * Pc, o.c_bproj);
                pack_mfma(p, [&](int k, int ch) {
                              const int st = 16 * (ch / 32) + (ch & 15);
                              return (ch & 16) ? bim(st, k) : bre(st, k);
                          }, H, 2 * Pc, o.c_bproj_pair);
                pack_mfma(p, [&](int k, int ch) { return idx[k] < 0 ? 0 : s.C_re[(size_t)ch * P + idx[k]]; }, Pc, H, o.c_cre);
                pack_mfma(p, [&](int k, int ch) { return idx[k] < 0 ? 0 : s.C_im[(size_t)ch * P + idx[k]]; }, Pc, H, o.c_cim);
                std::vector<int32_t> ar(Pc, 0), ai(Pc, 0);
                for (int st = 0; st < Pc; ++st)
                    if (idx[st] >= 0) {
                        ar[st] = s.A_re[idx[st]];
                        ai[st] = s.A_im[idx[st]];
                    }
                o.c_a_re = reinterpret_cast<const int32_t *>(put_raw(p, ar.data(), ar.size() * 4));
                o.c_a_im = reinterpret_cast<const int32_t *>(put_raw(p, ai.data(), ai.size() * 4));
            }
        }
        std::vector<int32_t> Dp(o.cre.w.Np, 0);
        for (int h = 0; h < H; ++h) Dp[h] = s.D[h];
        o.Dpad = reinterpret_cast<const int32_t *>(put_raw(p, Dp.data(), Dp.size() * 4));
        // gate operand r = change_cfg(sigmoid(xx)) for every (sign, min(|xx| >> sx, 6), |xx| mod 2^sx): fxpmodel.py:97-144
        // + :1075-1093, the same integer formula as fxp_prims.hpp sigmoid_lut
        const int sx = l.sig_x_exp, S = 1 << sx, sy = l.sig_y_exp;
        std::vector<int32_t> tab((size_t)14 * S);
        for (int pos = 0; pos < 2; ++pos)
            for (int i = 0; i < 7 * S; ++i) {
                const int ind = i >> sx, mu = i & (S - 1);
                const int32_t half = wadd(asr(wmul(S - mu, l.lut[ind]), sx), asr(wmul(mu, l.lut[ind + 1]), sx));
                const int32_t sg = wadd(1 << (sy - 1), pos ? half : wsub(0, half));
                tab[(size_t)pos * 7 * S + i] = chcfg(sg, l.out2.out_bits, sy, l.r_bits, l.r_exp);
            }
        o.sigtab = reinterpret_cast<const int32_t *>(put_raw(p, tab.data(), tab.size() * 4));
        // the sigmoid input is xx = gq >> (out_exp - sx): when that leaves <= 12 bits, r is tabulated over xx itself
        const int nb = l.out2.out_bits - (l.out2.out_exp - sx);
        if (l.out2.out_exp >= sx && nb >= 2 && nb <= SIGDIR_MAX_BITS && l.r_bits <= 16) {
            std::vector<int16_t> dir((size_t)1 << nb);
            for (int i = 0; i < (1 << nb); ++i) {
                const int32_t xx = i - (1 << (nb - 1)), ax = xx < 0 ? -xx : xx;
                const int ind = (ax >> sx) > 6 ? 6 : (ax >> sx), mu = ax & (S - 1);
                const int32_t half = wadd(asr(wmul(S - mu, l.lut[ind]), sx), asr(wmul(mu, l.lut[ind + 1]), sx));
                const int32_t sg = wadd(1 << (sy - 1), xx > 0 ? half : wsub(0, half));
                dir[i] = (int16_t)chcfg(sg, l.out2.out_bits, sy, l.r_bits, l.r_exp);
            }
            o.sigdir = reinterpret_cast<const int16_t *>(put_raw(p, dir.data(), dir.size() * 2));
            o.sigdir_bits = nb;
        }
    }
    const s5fxp_dense_desc &dd = d->decoder;
    pack_mfma(p, [&](int k, int ch) { return dd.weight[(size_t)k * dd.M + ch]; }, dd.K, dd.M, f->dec);
    pack_bias_eff(p, dd, f->dec.w.Np, f->dec);
}

struct FastWs {
    size_t hA, hB, x1, z, u, bq, xs, dyn, ext, dyn_bytes, total;
    int TB;
};

FastWs fast_ws(const s5fxp_model *m, int B, int L)
{
    const size_t N = (size_t)B * L;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    FastWs w{};
    w.TB = ((L + 3) / 4 + SCAN_DEPTH - 1) / SCAN_DEPTH * SCAN_DEPTH;
    size_t off = 0;
    const size_t nh = al(N * m->H * 2 + 64) + m->cfg.plane_skew; // int16 (+ slack for the 32-byte fragment loads of clamped tail lanes)
    const size_t ns = al(((size_t)B * w.TB + SCAN_DEPTH) * m->P * 8 * 4) + m->cfg.plane_skew;
    w.hA = off; off += nh;
    w.hB = off; off += nh;
    w.x1 = off; off += nh;
    w.z = off; off += nh;
    w.u = off; off += nh;
    w.bq = off; off += ns;
    w.xs = off; off += ns;
    // per-layer device state and per-channel extremes: contiguous, zeroed by one memset per forward
    w.dyn = off; off += al(sizeof(LayerDyn) * (size_t)m->n_layers);
    w.ext = off; off += al(sizeof(float) * 2 * (size_t)m->H * (size_t)m->n_layers * EXT_REPS);
    w.dyn_bytes = off - w.dyn;
    w.total = off;
    return w;
}

template <class K, class A>
inline void launch_smem(K kernel, unsigned grid, size_t smem, hipStream_t st, const A &args, unsigned threads, int G, const GroupOff &go)
{
    if (smem > 65536) // the dim_scale 1.0 tiles need more than the default dynamic-LDS limit
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kernel, dim3(grid, G), dim3(threads), smem, st, args, go);
}

// one workgroup = 4 waves x 32 frames; cap the grid at 4 workgroups per CU and let waves loop
inline unsigned mfma_grid(int64_t N)
{
    const int64_t blocks = ((N + 31) / 32 + 3) / 4;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks));
}

// Which recurrence kernel a fused forward with these S5FXP_FWD_* flags runs for layer li.  Codes 1..4 as
// s5fxp_model_recurrence_kernel (include/s5fxp.h); RK_EXACT = the exact 32-bit chain (k_scan_quad32_asm).  The one place
// that decides: forward_fast launches by it, the status words report it, the query answers from it.
enum { RK_LANE = 0, RK_QUAD32 = 1, RK_QUAD16 = 2, RK_PAIR = 3, RK_PAIRL = 4, RK_EXACT = 5 };
struct Rung {
    bool exact, defer, quad, s16, pair, pairl;
    int code;
};
Rung select_rung(const s5fxp_model *m, int li, int fwd_flags, bool traced, bool compact)
{
    const LayerDev &l = m->layers[li];
    const bool quad_ok = compact ? m->fast->layers[li].c_bounds.quad_ok : l.quad_ok;
    const bool pair_ok = compact ? m->fast->layers[li].c_bounds.pair_ok : l.pair_ok;
    const s5fxp_ssm_desc &s = l.sd;
    Rung r{};
    r.exact = (fwd_flags & S5FXP_FWD_EXACT) != 0;
    r.defer = (fwd_flags & S5FXP_FWD_DEFER_REDO) && !r.exact;
    r.quad = quad_ok && !r.exact;
    const int sh_re = s.Bu_re_exp - s.x_re_exp, sh_im = s.Bu_im_exp - s.x_im_exp;
    // optimistic forwards (the caller repeats with S5FXP_FWD_EXACT if the range check fires) keep both recurrence
    // streams as int16 when every Bu value, shifted to the state exponent, provably fits: half the bytes of the
    // B projection's output, of both sides of the recurrence and of the gate kernel's state input
    r.s16 = r.defer && r.quad && !traced && s.Bu_re_bits - sh_re <= 16 && s.Bu_im_bits - sh_im <= 16;
    // ... and, where the layer's coefficients leave room for Bu in the multiply's addend, the pair kernel (two lanes
    // per state, four instructions per step), fed either from an int16 Bu stream through LDS by a helper wave (default:
    // the HBM bytes of the quad16 path) or from an int32 K stream in global memory (ModelCfg::pair_global)
    r.pair = r.s16 && pair_ok && !m->cfg.no_pair && !(fwd_flags & S5FXP_FWD_NO_PAIR);
    r.pairl = r.pair && !m->cfg.pair_global;
    r.code = r.pairl ? RK_PAIRL : r.pair ? RK_PAIR : r.quad ? (r.s16 ? RK_QUAD16 : RK_QUAD32) : RK_EXACT;
    return r;
}

// State slots a compacted layer keeps in its two recurrence streams, or 0 = all of them: on the two int16 rungs (LDS-fed pair
// kernel, quad16) the padding slots are neither stored nor loaded -- whole state pairs, so that the pair kernel keeps whole
// lane quads (scan_quad.hpp ScanPairLArgs::live_slots; producers k_bproj_p<.., SM = 3 / 1>, consumer k_cgate_p<.., S16>).
int stream_live_slots(const s5fxp_model *m, int li, const Rung &rung, bool compact)
{
    const FastLayer &fl = m->fast->layers[li];
    const bool int16_rung = rung.pairl || (rung.quad && rung.s16 && !rung.pair);
    if (!(compact && int16_rung) || rung.exact || m->cfg.no_live_lanes) return 0;
    int n = 2 * ((fl.n_live + 1) / 2);
    n = n < 2 ? 2 : n;
    return n >= fl.c_slots ? 0 : n;
}

// G > 1: a grouped launch (include/s5fxp.h s5fxp_forward_opts::groups): x, y, workspace, status and the carry arrays hold G
// consecutive copies of what one forward uses; every kernel runs with gridDim.y = G (scan_quad.hpp GroupOff).  The caller
// (s5fxp_model_forward) sends only hook-free, trace-free forwards here with G > 1.
bool fast_bn_ext(const s5fxp_model *m)
{
    // BatchNorm exponents from per-channel extremes need every BN operand to be <= 16 bit with exponents in
    // [0,15] (no int32 wrap -> every stage monotone, mfma_bn.hpp); otherwise the four full reductions run.
    // ModelCfg::no_bn_ext (tests): take the four-reduction path even when the extremes method applies
    bool bn_ext = !m->cfg.no_bn_ext && m->enc.out_bits <= 16 && m->enc.out_exp >= 0 && m->enc.out_exp <= 15;
    for (int li = 0; li < m->n_layers; ++li) {
        const s5fxp_norm_desc &n = m->layers[li].nd;
        auto ok = [](int bits, int e) { return bits <= 16 && e >= 0 && e <= 15; };
        bn_ext = bn_ext && ok(n.mean_bits, n.mean_exp) && ok(n.invsq_var_bits, n.invsq_var_exp) &&
                 (!m->layers[li].scale || ok(n.scale_bits, n.scale_exp)) && (!m->layers[li].nbias || ok(n.bias_bits, n.bias_exp)) &&
                 m->layers[li].res_bits <= 16;
    }
    return bn_ext;
}

int forward_fast(const s5fxp_model *m, const int32_t *x, int x_bits, int x_exp, int B, int L, int32_t *y, void *workspace,
                 int32_t *status, const s5fxp_layer_trace *traces, const s5fxp_forward_opts *opts, hipStream_t st, int G = 1,
                 size_t ws_stride = 0)
{
    const FastModel &F = *m->fast;
    const ModelCfg &cfg = m->cfg;
    const int fwd_flags = opts ? opts->flags : 0;
    s5fxp_allreduce_max_fn allreduce = opts ? opts->allreduce : nullptr;
    void *allreduce_ctx = opts ? opts->allreduce_ctx : nullptr;
    void **scan_events = opts ? opts->scan_events : nullptr;
    void **gate_events = opts ? opts->gate_events : nullptr;
    const int32_t *state_in = opts ? opts->state_in : nullptr;
    int32_t *state_out = opts ? opts->state_out : nullptr;
    const bool exact = (fwd_flags & S5FXP_FWD_EXACT) != 0;
    const bool defer = (fwd_flags & S5FXP_FWD_DEFER_REDO) && !exact;
    const FastWs w = fast_ws(m, B, L);
    GroupOff go{};
    go.x = (int64_t)B * L * m->d_in * 4; go.y = (int64_t)B * L * m->d_out * 4; go.ws = (int64_t)(ws_stride ? ws_stride : w.total); go.status = 4 * S5FXP_STATUS_WORDS;
    go.state_in = go.state_out = (int64_t)m->n_layers * 2 * B * (m->P ? m->P : 1) * 4;
    // ModelCfg::debug_sync: synchronise and check for launch / execution errors after every stage (names the stage that
    // failed); off by default -- a forward has no host synchronisation, and launch errors are collected once at the end
    const bool debug_sync = cfg.debug_sync;
    auto stage_ok = [&](const char *what, int layer) -> bool {
        if (!debug_sync) return true;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) std::fprintf(stderr, "[s5fxp] %s (layer %d): %s\n", what, layer, hipGetErrorString(e));
        return e == hipSuccess;
    };
    char *ws = reinterpret_cast<char *>(workspace);
    auto I16 = [&](size_t off) { return reinterpret_cast<int16_t *>(ws + off); };
    auto I32 = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    LayerDyn *dyn = reinterpret_cast<LayerDyn *>(ws + w.dyn);
    const int64_t N = (int64_t)B * L;
    const int H = m->H, P = m->P;
    const int64_t NH = N * H;
    const bool big = (H == 192);
    const unsigned grid = mfma_grid(N);
    int rc;
    // the status words start from zero except what the host already knows: which path runs ([2]) and which recurrence
    // kernel each layer gets ([8 + 8l + 5])
    StatusInit si{};
    si.path = S5FXP_PATH_FUSED;
    for (int li = 0; li < m->n_layers; ++li) {
        const bool compact = F.layers[li].compact_ok && !cfg.no_compact && !traces && !state_in && !state_out;
        si.rk[li] = select_rung(m, li, fwd_flags, traces != nullptr, compact).code;
        si.slots[li] = compact ? F.layers[li].c_slots : m->P;
        const int ls = stream_live_slots(m, li, select_rung(m, li, fwd_flags, traces != nullptr, compact), compact);
        si.stream[li] = ls ? ls : si.slots[li];
    }
    hipLaunchKernelGGL(k_clear2, dim3(8, G), dim3(256), 0, st, status, (int)S5FXP_STATUS_WORDS, reinterpret_cast<int32_t *>(dyn),
                       (int)(w.dyn_bytes / 4), si, m->n_layers, go);
    const bool bn_ext = fast_bn_ext(m);

    // workgroups per launch (persistent loops over tiles): tuned per kernel on MI355X (256 CUs); ModelCfg (S5FXP_WGS_* at
    // model creation) overrides them
    // A grouped launch shares the caps between its groups (every workgroup pays its prologue once -- weights into registers,
    // tables into LDS, the exponent derivation -- and G x 512 workgroups of 4 tiles each pay it 4 times as often as 512
    // workgroups of 16 tiles: tools/sweep_groups.sh, +4 % at G = 4), down to a floor that still fills the chip with G groups
    auto per_group = [&](int64_t cap, int64_t floor_) { return G > 1 ? std::max<int64_t>(cap / G, floor_) : cap; };
    const int64_t cap_enc = per_group(cfg.cap_enc, 64), cap_dec = per_group(cfg.cap_dec, 64), cap_cgate = per_group(cfg.cap_cgate, 64),
                  cap_bproj = per_group(cfg.cap_bproj, 128), cap_resid = per_group(cfg.cap_resid, 64);
    // six-wave phase-split kernels (proj_p.hpp, mfma_fused.hpp): 64-frame tiles
    const int64_t tiles64 = (N + 63) / 64;
    auto grid_for = [&](int64_t tiles, int64_t cap) {
        const int64_t per = (tiles + cap - 1) / cap;
        return (unsigned)((tiles + per - 1) / per);
    };
    const unsigned grid_enc = grid_for(tiles64, cap_enc), grid_dec = grid_for(tiles64, cap_dec);
    auto launch6g = [&](auto kernel, unsigned g6, size_t smem, const auto &args, unsigned threads = 384) {
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kernel, dim3(g6, G), dim3(threads), smem, st, args, go);
    };
    auto launch6 = [&](auto kernel, size_t smem, const auto &args) { launch6g(kernel, grid_dec, smem, args); };
    // the gate kernel's launch can carry a pair of HIP events (s5fxp_forward_opts::gate_events: measurement only)
    hipEvent_t gev0 = nullptr, gev1 = nullptr;
    auto launch_gate = [&](auto kernel, unsigned g6, size_t smem, const auto &args, unsigned threads = 384) {
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (gev0 && gev1) hipExtLaunchKernelGGL(kernel, dim3(g6, G), dim3(threads), smem, st, gev0, gev1, 0, args, go);
        else hipLaunchKernelGGL(kernel, dim3(g6, G), dim3(threads), smem, st, args, go);
    };
    auto launch6x = [&](auto kernel, size_t smem, const auto &args, float *ext, int ext_reps) {
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kernel, dim3(grid_enc, G), dim3(384), smem, st, args, ext, ext_reps, go);
    };

    // residual / extremes pass: a workgroup owns rm_span consecutive frames, a multiple of its 4 x R frame step
    const int64_t rm_step = 4 * (RESID_THREADS / (H / 8)), rm_iters = (N + rm_step - 1) / rm_step;
    // (the kernel addresses a workgroup's span with 32-bit byte offsets: at most 2^31 bytes of one (N,H) int16 tensor per workgroup)
    const int64_t rm_per_max = std::max<int64_t>(1, ((int64_t(1) << 31) / (2 * H)) / rm_step);
    const int64_t rm_per = std::min(rm_per_max, (rm_iters + cap_resid - 1) / cap_resid), rm_span = rm_per * rm_step;
    const unsigned rm_grid = (unsigned)((rm_iters + rm_per - 1) / rm_per);

    int16_t *h = I16(w.hA), *hn = I16(w.hB);
    int hb = m->enc.out_bits;
    DynExp he{m->enc.out_exp, nullptr};
    // BatchNorm arguments of layer li, whose input has hb_in bits and the (device) exponent he_in
    auto make_bn = [&](int li, int hb_in, DynExp he_in) {
        const LayerDev &l = m->layers[li];
        const s5fxp_ssm_desc &s = l.sd;
        auto mx = [](int a, int b) { return a > b ? a : b; };
        BnArgs bn{};
        bn.mm = l.mm; bn.isv = l.isv; bn.scale = l.scale; bn.bias = l.nbias;
        bn.xb = hb_in; bn.xe = he_in;
        bn.mb = l.nd.mean_bits; bn.me = l.nd.mean_exp; bn.b1 = mx(hb_in, bn.mb);
        bn.ib = l.nd.invsq_var_bits; bn.ie = l.nd.invsq_var_exp; bn.b2 = mx(bn.b1, bn.ib);
        bn.sb = l.nd.scale_bits; bn.se = l.nd.scale_exp; bn.b3 = l.scale ? mx(bn.b2, bn.sb) : bn.b2;
        bn.bb = l.nd.bias_bits; bn.be = l.nd.bias_exp; bn.b4 = l.nbias ? mx(bn.b3, bn.bb) : bn.b3;
        bn.ub = s.u_bits; bn.ue = s.u_exp; bn.out_bits = bn.b4; bn.dyn = dyn + li;
        return bn;
    };
    // ---- encoder + ReLU
    {
        const DenseDev &e = m->enc;
        EncArgs a{};
        a.x = x; a.y = h; a.w = F.enc.w; a.bias_eff = F.enc.bias_eff; a.N = N; a.K = e.K; a.M = e.M;
        a.xb = x_bits; a.xe = x_exp; a.inp_bits = e.inp_bits; a.inp_exp = e.inp_exp;
        a.conv = (x_bits > e.inp_bits || x_exp > e.inp_exp) ? 1 : 0;
        a.rs = (a.conv ? e.inp_exp : x_exp) + e.w_exp - e.out_exp;
        if (!shift_ok(a.rs)) return S5FXP_ENEGSHIFT;
        a.out_bits = e.out_bits; a.status = status;
        const size_t smem = 4 * (size_t)H * 4 + 2 * 64 * 304; // cs128 + bias_eff + byte planes + extremes
        // the extremes of the output (layer 0's BatchNorm operand) are gathered on the way; single-rank mode also
        // lets the last workgroup derive layer 0's BatchNorm exponents (mfma_bn.hpp ResidTail)
        float *ext0 = bn_ext && m->n_layers > 0 ? reinterpret_cast<float *>(ws + w.ext) : nullptr;
        const int ext_reps = (ext0 && !allreduce) ? EXT_REPS : 1; // the consumer (k_bproj_p's prologue) folds the replicas
        if (big) launch6x(k_enc_p<6>, smem, a, ext0, ext_reps);
        else launch6x(k_enc_p<3>, smem, a, ext0, ext_reps);
        if (!stage_ok("encoder", -1)) return S5FXP_EHIP;
    }
    // single-rank mode folds the two one-workgroup "finalize" kernels of every layer into the residual pass
    // (mfma_bn.hpp k_resid_minmax16); with a multi-rank hook the maxima are exchanged in between, so they stay
    const bool fold = bn_ext && !allreduce;

    bool dec_resid = false; // the decoder does the last layer's residual pass itself
    DecResid dz{};
    int dec_bits = 0;
    for (int li = 0; li < m->n_layers; ++li) {
        const LayerDev &l = m->layers[li];
        const FastLayer &fl = F.layers[li];
        const s5fxp_layer_trace *tr = traces ? &traces[li] : nullptr;
        LayerDyn *d = dyn + li;
        int32_t *st_exps = status + 8 + 8 * li;
        const s5fxp_ssm_desc &s = l.sd;
        auto mx = [](int a, int b) { return a > b ? a : b; };

        const BnArgs bn = make_bn(li, hb, he);

        const unsigned rg = ew_grid(NH / 4) > 2048 ? 2048 : ew_grid(NH / 4);
        auto hook = [&](int slot, int n) -> int {
            return allreduce ? allreduce(allreduce_ctx, reinterpret_cast<float *>(d->mx + slot), n, (void *)st) : 0;
        };
        if (bn_ext) {
            float *ext = reinterpret_cast<float *>(ws + w.ext) + (size_t)li * 2 * H * EXT_REPS;
            // layer 0: extremes of the encoder output; later layers: the previous layer's residual pass left them
            // layer 0: the encoder left the extremes; later layers: the previous layer's residual pass
            if (!fold) {
                // mode A: the extremes (positive floats) are what the ranks exchange -- one MAX over 2H values
                if (allreduce && allreduce(allreduce_ctx, ext, 2 * H, (void *)st)) return S5FXP_EHIP;
                hipLaunchKernelGGL(k_bn_finalize_mm, dim3(1), dim3(256), 0, st, bn, (const float *)ext, H, d, status, st_exps);
            }
        } else {
        hipLaunchKernelGGL(k_bn_reduce16<1>, dim3(rg), dim3(256), 0, st, bn, (const int16_t *)h, NH, H, d);
        if (hook(0, 3)) return S5FXP_EHIP;
        hipLaunchKernelGGL(k_bn_finalize<1>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        hipLaunchKernelGGL(k_bn_reduce16<2>, dim3(rg), dim3(256), 0, st, bn, (const int16_t *)h, NH, H, d);
        if (hook(3, 1)) return S5FXP_EHIP;
        hipLaunchKernelGGL(k_bn_finalize<2>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        if (l.scale) {
            hipLaunchKernelGGL(k_bn_reduce16<3>, dim3(rg), dim3(256), 0, st, bn, (const int16_t *)h, NH, H, d);
            if (hook(4, 1)) return S5FXP_EHIP;
            hipLaunchKernelGGL(k_bn_finalize<3>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        }
        if (l.nbias) {
            hipLaunchKernelGGL(k_bn_reduce16<4>, dim3(rg), dim3(256), 0, st, bn, (const int16_t *)h, NH, H, d);
            if (hook(5, 3)) return S5FXP_EHIP;
            hipLaunchKernelGGL(k_bn_finalize<4>, dim3(1), dim3(64), 0, st, bn, d, status, st_exps);
        }
        }

        // ---- B projection -> scan-native stream (+ u for the C projection)
        const int sh_re = s.Bu_re_exp - s.x_re_exp, sh_im = s.Bu_im_exp - s.x_im_exp;
        // live-state compaction (FastLayer): this layer's kernels run on c_slots state slots
        const bool compact = fl.compact_ok && !cfg.no_compact && !tr && !state_in && !state_out;
        const Rung rung = select_rung(m, li, fwd_flags, tr != nullptr, compact);
        const bool quad = rung.quad, s16 = rung.s16, pair = rung.pair, pairl = rung.pairl;
        const int32_t l_pair_xmax = compact ? fl.c_bounds.pair_xmax : l.pair_xmax, l_quad_xmax = compact ? fl.c_bounds.quad_xmax : l.quad_xmax;
        const int P = compact ? fl.c_slots : m->P;
        // on the LDS-fed pair rung a compacted layer with ONE group of 32 slots keeps only its live slots in the two recurrence
        // streams (whole lane quads = state pairs; scan_quad.hpp ScanPairLArgs::live_lanes): 0 = every slot
        const int live_slots = stream_live_slots(m, li, rung, compact);
        const int32_t *la_re = compact ? fl.c_a_re : l.a_re, *la_im = compact ? fl.c_a_im : l.a_im;
        const MfmaW &w_bproj = compact ? fl.c_bproj.w : fl.bproj.w, &w_bproj_pair = compact ? fl.c_bproj_pair.w : fl.bproj_pair.w;
        const MfmaW &w_cre = compact ? fl.c_cre.w : fl.cre.w, &w_cim = compact ? fl.c_cim.w : fl.cim.w;
        // packed int16 epilogues of the gate kernel (mfma_fused.hpp PK16): every width they touch is 16, no out2 input conversion
        const bool out2_conv = s.y_bits > l.out2.inp_bits || s.y_exp > l.out2.inp_exp;
        const bool direct = s16 && fl.sigdir_bits > 0;
        const bool pk16 = direct && !tr && !cfg.no_pk16 && fl.bias16 && s.y_bits == 16 && l.out2.out_bits == 16 && l.res_bits == 16 &&
                          l.l_bits == 16 && !out2_conv && l.l_exp - s.y_exp <= 14;
        // ... and on that kernel the SSM input u CAN be recomputed from the layer input instead of travelling through memory
        // (k_cgate_p<.., GBN>, S5FXP_GATE_BN=1; the exponents are the ones this layer's B projection publishes in its prologue).
        // Off by default: -50 MB per layer and batch, but the twelve VALU operations per element land in the kernel that is
        // VALU-co-limited already -- gate kernel 207 -> 250 us per 8-batch launch, B projection 104 -> 90: 3 % slower overall
        const bool gate_bn = pk16 && fold && !big && cfg.gate_bn;
        {
            BprojM2Args a{};
            a.bn = bn; a.x = h; a.w = w_bproj; a.bq = I32(w.bq); a.u = I16(w.u);
            a.tr_bu_re = tr ? tr->Bu_re : nullptr; a.tr_bu_im = tr ? tr->Bu_im : nullptr;
            a.tr_pre_s5 = tr ? tr->pre_s5 : nullptr; a.tr_u = tr ? tr->u : nullptr;
            a.N = N; a.L = L; a.TB = w.TB; a.H = H; a.P = P;
            a.rs_re = s.u_exp + s.B_re_exp - s.Bu_re_exp; a.rs_im = s.u_exp + s.B_im_exp - s.Bu_im_exp;
            a.bre_bits = s.Bu_re_bits; a.bim_bits = s.Bu_im_bits; a.sh_re = sh_re; a.sh_im = sh_im;
            a.k_re = 65536 - (1 << (16 - s.A_re_exp));
            a.live_slots = live_slots;
            a.no_u = gate_bn ? 1 : 0;
            if (fold) {
                a.ext = reinterpret_cast<const float *>(ws + w.ext) + (size_t)li * 2 * H * EXT_REPS;
                a.ext_reps = EXT_REPS; a.status = status; a.status_exps = st_exps;
            }
            // phase-split kernel (proj_p.hpp): 64-step tiles, up to 4 (H=96) / 2 (H=192) workgroups per CU
#ifdef S5_BPROJ_CSR
            const size_t smem = 16 * (size_t)H + 4 * 64 * (size_t)(H + 16) + 2 * (size_t)(2 * m->P) * S5_BPROJ_CSR + 64; // + compressed columns
#else
            const size_t smem = 16 * (size_t)H + 4 * 64 * (size_t)(H + 16); // BN operands + double-buffered byte planes
#endif
            {
                a.t_lo = 0; a.t_len = L;
                const int64_t tl = (int64_t)B * ((a.t_len + 63) / 64), cap = big ? cap_bproj / 2 : cap_bproj, per = (tl + cap - 1) / cap;
                const unsigned bthr = big ? 512 : 256; // one wave per 32-column tile of [B_re | B_im]
                const unsigned pgrid = (unsigned)((tl + per - 1) / per);
                // SM: the stream the recurrence rung wants (proj_p.hpp); a compacted layer has half the column tiles
                auto bproj = [&](auto sm) {
                    constexpr int SM = decltype(sm)::value;
                    if (big) {
                        if (compact && P == 32) launch_smem(k_bproj_p<6, 8, false, SM, 2>, pgrid, smem, st, a, bthr, G, go);
                        else if (compact) launch_smem(k_bproj_p<6, 8, false, SM, 4>, pgrid, smem, st, a, bthr, G, go);
                        else launch_smem(k_bproj_p<6, 8, false, SM, 8>, pgrid, smem, st, a, bthr, G, go);
                    } else {
                        if (compact) launch_smem(k_bproj_p<3, 4, false, SM, 2>, pgrid, smem, st, a, bthr, G, go);
                        else launch_smem(k_bproj_p<3, 4, false, SM, 4>, pgrid, smem, st, a, bthr, G, go);
                    }
                };
                if (tr) {
                    if (big) launch_smem(k_bproj_p<6, 8, true>, pgrid, smem, st, a, bthr, G, go);
                    else launch_smem(k_bproj_p<3, 4, true>, pgrid, smem, st, a, bthr, G, go);
                } else if (pairl) {
                    a.w = w_bproj_pair;
                    bproj(std::integral_constant<int, 3>{});
                } else if (pair) {
                    a.w = w_bproj_pair;
                    bproj(std::integral_constant<int, 2>{});
                } else if (s16) {
                    bproj(std::integral_constant<int, 1>{});
                } else {
                    bproj(std::integral_constant<int, 0>{});
                }
            }
        }
        if (!stage_ok("B projection", li)) return S5FXP_EHIP;
        // ---- recurrence
        const size_t plane = (size_t)B * P;
        const int32_t *x0_re = state_in ? state_in + (size_t)li * 2 * plane : nullptr, *x0_im = state_in ? x0_re + plane : nullptr;
        int32_t xmax = 32767; // the C projection's 16-bit planes
        hipStream_t sst = st; // the stream the recurrence runs on
        // measurement: the two events are attached to the launch itself (start / stop time stamps of this dispatch,
        // what rocprofv3's kernel trace reports), not recorded around it
        hipEvent_t ev0 = scan_events ? (hipEvent_t)scan_events[2 * li] : nullptr, ev1 = scan_events ? (hipEvent_t)scan_events[2 * li + 1] : nullptr;
        auto launch_scan = [&](auto kernel, dim3 grid, dim3 block, auto args) {
            grid.y = G;
            if (ev0 && ev1) hipExtLaunchKernelGGL(kernel, grid, block, 0, sst, ev0, ev1, 0, args, go);
            else hipLaunchKernelGGL(kernel, grid, block, 0, sst, args, go);
        };
        if (pairl) {
            ScanPairLArgs q{};
            q.b16 = I16(w.bq); q.xs = I16(w.xs); q.a_re = la_re; q.a_im = la_im; q.B = B; q.TB = w.TB; q.P = P;
            q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.x0_re = x0_re; q.x0_im = x0_im; q.live_slots = live_slots;
            // one helper wave (a second one lands on the computing wave's side of the LDS path and costs more than it
            // helps: profiles/r02_ubench_pair.log).  Blocks per LDS buffer = steps per s_barrier / 4: S5FXP_PAIRL_BLOCKS
            const int blocks = cfg.pairl_blocks;
            const dim3 sgrid((unsigned)((int64_t)B * (P / 32)), G);
            auto launch_pairl = [&](auto kernel, int smem_bytes) {
                if (smem_bytes > 65536) // > 64 KB of dynamic LDS needs the attribute (idempotent, a host-side table update)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
                if (ev0 && ev1) hipExtLaunchKernelGGL(kernel, sgrid, dim3(128), smem_bytes, sst, ev0, ev1, 0, q, go);
                else hipLaunchKernelGGL(kernel, sgrid, dim3(128), smem_bytes, sst, q, go);
            };
            if (blocks == 16) launch_pairl(k_scan_pairl_asm<16>, 3 * 16 * 1024);
            else launch_pairl(k_scan_pairl_asm<32>, 3 * 32 * 1024);
            xmax = l_pair_xmax < xmax ? l_pair_xmax : xmax;
        } else if (pair) {
            ScanPairArgs q{};
            q.k = I32(w.bq); q.xs = I16(w.xs); q.a_re = la_re; q.a_im = la_im; q.B = B; q.TB = w.TB; q.P = P;
            q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.x0_re = x0_re; q.x0_im = x0_im;
            launch_scan(k_scan_pair_asm, dim3((unsigned)((int64_t)B * (P / 32))), dim3(64), q);
            xmax = l_pair_xmax < xmax ? l_pair_xmax : xmax; // <= 32766: a saturated int16 state fails the check
        } else if (quad) {
            ScanQuadArgs q{};
            q.bq = I32(w.bq); q.xs = I32(w.xs); q.a_re = la_re; q.a_im = la_im; q.B = B; q.TB = w.TB; q.P = P;
            q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.x0_re = x0_re; q.x0_im = x0_im; q.live_slots = live_slots;
            {
                if (s16) launch_scan(k_scan_quad_asm16, dim3((unsigned)((int64_t)B * P / 16)), dim3(64), q);
                else launch_scan(k_scan_quad_asm, dim3((unsigned)((int64_t)B * P / 16)), dim3(64), q);
            }
            xmax = l_quad_xmax < xmax ? l_quad_xmax : xmax;
            if (s16 && xmax > 32766) xmax = 32766; // a saturated int16 state must fail the check
        } else {
            // states of any width: the exact 32-bit chain in the same quad layout
            ScanQuadArgs q{};
            q.bq = I32(w.bq); q.xs = I32(w.xs); q.a_re = la_re; q.a_im = la_im; q.B = B; q.TB = w.TB; q.P = P;
            q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.run_if = nullptr; q.x0_re = x0_re; q.x0_im = x0_im;
            launch_scan(k_scan_quad32_asm, dim3((unsigned)((int64_t)B * P / 16)), dim3(64), q);
        }
        if (!stage_ok("recurrence", li)) return S5FXP_EHIP;
        // ---- fused C projection + D*u + ReLU + out2 + sigmoid + gate (+ range check, + residual maxima)
        GateMArgs ga{};
        bool fused = false;
        {
            const DenseDev &o = l.out2;
            ga.x1 = I16(w.x1); ga.skip = h; ga.z = I16(w.z); ga.w = fl.out2.w; ga.bias_eff = fl.out2.bias_eff;
            ga.tr_out2 = tr ? tr->out2 : nullptr; ga.tr_sig = tr ? tr->out2_sigmoid : nullptr;
            ga.tr_z = tr ? tr->post_GLU : nullptr;
            ga.N = N; ga.H = H; ga.y_bits = s.y_bits; ga.y_exp = s.y_exp;
            ga.conv = (s.y_bits > o.inp_bits || s.y_exp > o.inp_exp) ? 1 : 0;
            ga.inp_bits = o.inp_bits; ga.inp_exp = o.inp_exp;
            ga.rs = (ga.conv ? o.inp_exp : s.y_exp) + o.w_exp - o.out_exp;
            if (!shift_ok(ga.rs)) return S5FXP_ENEGSHIFT;
            ga.out_bits = o.out_bits; ga.out_exp = o.out_exp; ga.sig_x = l.sig_x; ga.sig_y = l.sig_y;
            std::memcpy(ga.lut, l.lut, sizeof(ga.lut));
            ga.l_bits = l.l_bits; ga.l_exp = l.l_exp; ga.r_bits = l.r_bits; ga.r_exp = l.r_exp; ga.res_bits = l.res_bits;
            ga.res_exp = l.res_exp; ga.rs_gate = l.l_exp + l.r_exp - l.res_exp; ga.skip_e = he; ga.dynw = d;
        }
        {
            CGateArgs a{};
            a.u = I16(w.u); a.skip = h; a.xs = I32(w.xs); a.w_re = w_cre; a.w_im = w_cim; a.w_o2 = fl.out2.w;
            a.D = fl.Dpad; a.bias_eff = fl.out2.bias_eff; a.z = I16(w.z); a.sigtab = fl.sigtab; a.sigdir = fl.sigdir; a.sigdir_bits = fl.sigdir_bits; a.mx_slot = 8;
            a.tr_ys = tr ? tr->ys : nullptr; a.tr_out2 = ga.tr_out2; a.tr_sig = ga.tr_sig; a.tr_z = ga.tr_z;
            a.N = N; a.L = L; a.TB = w.TB; a.H = H;
            a.rs_re = s.x_re_exp + s.C_re_exp - s.y_exp; a.rs_im = s.x_im_exp + s.C_im_exp - s.y_exp;
            a.rs_d = s.D_exp + s.u_exp - s.y_exp; a.y_bits = s.y_bits; a.y_exp = s.y_exp; a.xmax = xmax;
            a.conv = ga.conv; a.inp_bits = ga.inp_bits; a.inp_exp = ga.inp_exp; a.rs_o2 = ga.rs; a.out_bits = ga.out_bits;
            a.out_exp = ga.out_exp; a.sig_x = l.sig_x; a.sig_y = l.sig_y;
            std::memcpy(a.lut, l.lut, sizeof(a.lut));
            a.l_bits = l.l_bits; a.l_exp = l.l_exp; a.r_bits = l.r_bits; a.r_exp = l.r_exp; a.res_bits = l.res_bits;
            a.res_exp = l.res_exp; a.rs_gate = ga.rs_gate; a.skip_e = he; a.dynw = d; a.status = status;
            a.live_slots = live_slots;
            // phase-split fused kernel (mfma_fused.hpp): six waves per workgroup, 64-frame tiles, no weights in LDS
            fused = true;
            a.bad_bits = ST_WIDE_STATE | (defer ? ST_REDO : 0);
            a.bn = bn;
            const size_t smem = 5 * (size_t)H * 4 + 32 + (direct ? (size_t)sigdir_lds_bytes(fl.sigdir_bits) : 4 * (size_t)SIGTAB_WORDS) +
                                2 * 64 * (size_t)(2 * P + 16) + 2 * 64 * (size_t)(H + 16) + 192 + (gate_bn ? 16 * (size_t)H : 0) +
                                (pk16 && !gate_bn ? 2 * 64 * (size_t)(2 * H + 8) : 0); // + the u / skip / z tiles (mfma_fused.hpp COAL)
            if (exact) {
                // S5FXP_FWD_EXACT: the exact kernels below are the only ones; raise their gate
                for (int g = 0; g < G; ++g)
                    if ((rc = hip_rc(hipMemsetAsync(reinterpret_cast<char *>(&d->redo) + (size_t)g * go.ws, 0xff, 4, st)))) return rc;
            } else {
                {
                    a.t_lo = 0; a.t_len = L;
                    const int64_t tl = (int64_t)B * ((a.t_len + 63) / 64), per = (tl + cap_cgate - 1) / cap_cgate;
                    const unsigned cg = (unsigned)((tl + per - 1) / per);
                    // <S16, DIRECT, PAIR, PK16> of mfma_fused.hpp; KS = state slots / 32 (halved for a compacted layer)
                    gev0 = gate_events ? (hipEvent_t)gate_events[2 * li] : nullptr; gev1 = gate_events ? (hipEvent_t)gate_events[2 * li + 1] : nullptr;
                    auto cgate = [&](auto s16_t, auto direct_t, auto pair_t, auto pk16_t) {
                        constexpr bool S16_ = decltype(s16_t)::value, DIR_ = decltype(direct_t)::value, PAIR_ = decltype(pair_t)::value,
                                       PK_ = decltype(pk16_t)::value;
                        if (big) {
                            if (compact && P == 32) launch_gate(k_cgate_p<1, 6, false, S16_, DIR_, 64, false, PAIR_, PK_>, cg, smem, a, 768);
                            else if (compact) launch_gate(k_cgate_p<2, 6, false, S16_, DIR_, 64, false, PAIR_, PK_>, cg, smem, a, 768);
                            else launch_gate(k_cgate_p<4, 6, false, S16_, DIR_, 64, false, PAIR_, PK_>, cg, smem, a, 768);
                        } else if (PK_ && !cfg.cgate_ft64 && !gate_bn) {
                            // 32-frame tiles, three-wave workgroups: with the sigmoid table sized exactly FIVE of them fit a CU's LDS and
                            // registers -- 15 waves instead of the 12 of two six-wave workgroups, for a kernel whose waves wait two thirds
                            // of their cycles.  The grid is exactly what is resident at once (5 x 256 CUs): 170 us per 8-batch launch
                            // against 190 (tools/ab_cgate_ft32.sh; 1024 or 1536 workgroups: 189 / 209).
                            const size_t smem32 = 5 * (size_t)H * 4 + 32 + (direct ? (size_t)sigdir_lds_bytes(fl.sigdir_bits) : 4 * (size_t)SIGTAB_WORDS) +
                                                  2 * 32 * (size_t)(2 * P + 16) + 2 * 32 * (size_t)(H + 16) + 192 + 2 * 32 * (size_t)(2 * H + 8);
                            const int64_t tl32 = (int64_t)B * ((L + 31) / 32), cap32 = std::max<int64_t>(cfg.cap_cgate32 / G, 1), per32 = (tl32 + cap32 - 1) / cap32;
                            const unsigned g32 = (unsigned)((tl32 + per32 - 1) / per32);
                            if (compact) launch_gate(k_cgate_p<1, 3, false, S16_, DIR_, 32, false, PAIR_, PK_>, g32, smem32, a, 192);
                            else launch_gate(k_cgate_p<2, 3, false, S16_, DIR_, 32, false, PAIR_, PK_>, g32, smem32, a, 192);
                        } else if (PK_ && gate_bn) {
                            if (compact) launch_gate(k_cgate_p<1, 3, false, S16_, DIR_, 64, false, PAIR_, PK_, PK_>, cg, smem, a);
                            else launch_gate(k_cgate_p<2, 3, false, S16_, DIR_, 64, false, PAIR_, PK_, PK_>, cg, smem, a);
                        } else {
                            if (compact) launch_gate(k_cgate_p<1, 3, false, S16_, DIR_, 64, false, PAIR_, PK_>, cg, smem, a);
                            else launch_gate(k_cgate_p<2, 3, false, S16_, DIR_, 64, false, PAIR_, PK_>, cg, smem, a);
                        }
                    };
                    using T_ = std::true_type;
                    using F_ = std::false_type;
                    if (tr) {
                        if (big) launch6g(k_cgate_p<4, 6, true>, cg, smem, a, 768);
                        else launch6g(k_cgate_p<2, 3, true>, cg, smem, a);
                    } else if (pk16 && pair) cgate(T_{}, T_{}, T_{}, T_{});
                    else if (pk16) cgate(T_{}, T_{}, F_{}, T_{});
                    else if (direct && pair) cgate(T_{}, T_{}, T_{}, F_{});
                    else if (pair) cgate(T_{}, F_{}, T_{}, F_{});
                    else if (direct) cgate(T_{}, T_{}, F_{}, F_{}); // (32-frame tiles with three-wave workgroups were tried: 39 vs 36 us)
                    else if (s16) cgate(T_{}, F_{}, F_{}, F_{});
                    else cgate(F_{}, F_{}, F_{}, F_{});
                }
            }
            // ---- exact re-run, only if a state left the fast kernels' range (LayerDyn::redo); with
            // S5FXP_FWD_DEFER_REDO the caller repeats the forward instead (S5FXP_ST_REDO)
            if (!defer) {
                if (quad) {
                    ScanQuadArgs q{};
                    q.bq = I32(w.bq); q.xs = I32(w.xs); q.a_re = la_re; q.a_im = la_im; q.B = B; q.TB = w.TB; q.P = P;
                    q.ea_re = s.A_re_exp; q.ea_im = s.A_im_exp; q.run_if = &d->redo; q.x0_re = x0_re; q.x0_im = x0_im;
                    hipLaunchKernelGGL(k_scan_quad32_asm, dim3((unsigned)((int64_t)B * P / 16), G), dim3(64), 0, st, q, go);
                }
                // the exact gate kernel: four byte planes of the int32 states, no range assumption; its maxima go to
                // slots 11..13, which the residual pass picks when `redo` is set
                CGateArgs e = a;
                e.run_if = &d->redo; e.mx_slot = 11; e.t_lo = 0; e.t_len = L; e.bad_bits = 0;
                const size_t smem_w = 5 * (size_t)H * 4 + 32 + 4 * (size_t)SIGTAB_WORDS + 4 * 64 * (size_t)(2 * P + 16) +
                                      2 * 64 * (size_t)(H + 16) + 192;
                const int64_t tlw = (int64_t)B * ((L + 63) / 64), perw = (tlw + 511) / 512;
                const unsigned cgw = (unsigned)((tlw + perw - 1) / perw);
                if (tr) {
                    if (big) launch6g(k_cgate_p<4, 6, true, false, false, 64, true>, cgw, smem_w, e, 768);
                    else launch6g(k_cgate_p<2, 3, true, false, false, 64, true>, cgw, smem_w, e);
                } else if (big) {
                    if (compact && P == 32) launch6g(k_cgate_p<1, 6, false, false, false, 64, true>, cgw, smem_w, e, 768);
                    else if (compact) launch6g(k_cgate_p<2, 6, false, false, false, 64, true>, cgw, smem_w, e, 768);
                    else launch6g(k_cgate_p<4, 6, false, false, false, 64, true>, cgw, smem_w, e, 768);
                } else {
                    if (compact) launch6g(k_cgate_p<1, 3, false, false, false, 64, true>, cgw, smem_w, e);
                    else launch6g(k_cgate_p<2, 3, false, false, false, 64, true>, cgw, smem_w, e);
                }
            }
            if (state_out) // carry out: the state after frame L-1, from whichever kernel wrote the stream last
                hipLaunchKernelGGL(k_state_out, dim3((unsigned)((plane + 255) / 256), G), dim3(256), 0, st, (const void *)I32(w.xs),
                                   pair ? 2 : (s16 ? 1 : 0), defer ? (const int32_t *)nullptr : (const int32_t *)&d->redo, B, L, P,
                                   w.TB, state_out + (size_t)li * 2 * plane, state_out + (size_t)li * 2 * plane + plane, go);
            if (tr && (tr->xs_re || tr->xs_im))
                hipLaunchKernelGGL(k_unpack_native, dim3(ew_grid(N * P)), dim3(256), 0, st, (const int32_t *)I32(w.xs),
                                   tr->xs_re, tr->xs_im, B, L, P, w.TB);
        }
        if (!stage_ok("gate kernel", li)) return S5FXP_EHIP;
        if (allreduce) {
            // ranks may differ in `redo`: move the valid maxima to slots 8..10 before they are exchanged
            if (fused) hipLaunchKernelGGL(k_select_maxima, dim3(1), dim3(64), 0, st, d);
            if (hook(8, 3)) return S5FXP_EHIP;
        }
        const int redo_slot = (allreduce || defer) ? 8 : 11; // mode A moved them; deferred: no re-run happened
        if (!fold)
            hipLaunchKernelGGL(k_res_finalize, dim3(1), dim3(64), 0, st, d, l.res_exp, he, l.res_bits, status, st_exps, redo_slot);
        const bool more_layers = li + 1 < m->n_layers;
        if (bn_ext && fold && !tr && !more_layers && !cfg.no_dec_resid) {
            // the last layer's residual pass rides on the decoder (proj_p.hpp k_dec_p<.., RESID>)
            dec_resid = true;
            dz.z = I16(w.z); dz.res_bits = l.res_bits; dz.skip_bits = hb;
            dz.hd.d = d; dz.hd.res_exp = l.res_exp; dz.hd.skip_e = he; dz.hd.redo_slot = redo_slot; dz.hd.status_exps = st_exps;
            dz.hd.enable = 1;
            dec_bits = l.res_bits;
            break;
        }
        if (bn_ext) {
            const bool more = li + 1 < m->n_layers;
            float *ext_next = more ? reinterpret_cast<float *>(ws + w.ext) + (size_t)(li + 1) * 2 * H * EXT_REPS : nullptr;
            ResidHead hd{};
            hd.d = d; hd.res_exp = l.res_exp; hd.skip_e = he; hd.redo_slot = redo_slot; hd.status_exps = st_exps;
            hd.enable = fold ? 1 : 0;
            const int ext_reps = (more && fold) ? EXT_REPS : 1; // the next layer's B projection derives its exponents from the extremes
            hipLaunchKernelGGL(k_resid_minmax16<true>, dim3(rm_grid, G), dim3(RESID_THREADS), 0, st, (const int16_t *)I16(w.z),
                               (const int16_t *)h, hn, tr ? tr->residadd : nullptr, N, H, rm_span, l.res_bits, hb, hd, ext_next, ext_reps,
                               status, go);
        } else {
            hipLaunchKernelGGL(k_resid16, dim3(ew_grid(NH / 4)), dim3(256), 0, st, (const int16_t *)I16(w.z),
                               (const int16_t *)h, hn, tr ? tr->residadd : nullptr, NH, l.res_bits, hb, (const LayerDyn *)d);
        }
        if (!stage_ok("residual pass", li)) return S5FXP_EHIP;
        int16_t *sw = h; h = hn; hn = sw;
        hb = l.res_bits;
        he = DynExp{0, &d->res.eo};
    }
    // ---- decoder
    {
        const DenseDev &e = m->dec;
        DecArgs a{};
        a.x = h; a.y = y; a.w = F.dec.w; a.bias_eff = F.dec.bias_eff; a.N = N; a.H = H; a.M = e.M;
        a.xb = dec_resid ? dec_bits : hb; a.xe = he; a.inp_bits = e.inp_bits; a.inp_exp = e.inp_exp; a.w_exp = e.w_exp;
        a.out_bits = e.out_bits; a.out_exp = e.out_exp; a.status = status;
        const size_t smem = 2 * 64 * (size_t)(H + 16);
        auto launch_dec = [&](auto kernel) {
            if (smem > 65536)
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(kernel, dim3(grid_dec, G), dim3(384), smem, st, a, dz, go);
        };
        if (dec_resid) {
            if (big) launch_dec(k_dec_p<6, true>); // 192 channels: 2 x 4 vectors of prefetch, one workgroup per CU
            else launch_dec(k_dec_p<3, true>);
        } else {
            if (big) launch_dec(k_dec_p<6, false>);
            else launch_dec(k_dec_p<3, false>);
        }
        if (!stage_ok("decoder", -1)) return S5FXP_EHIP;
    }
    return launch_rc();
}

} // namespace
