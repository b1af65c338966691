// mfma_fused.hpp -- C projection, D*u, ReLU, out2 dense, LUT sigmoid and mult_gate in ONE kernel.
//
// fxpmodel.py:740-793 (C projection, x2, + D*u), :1125 (ReLU), :1133-1137 (out2, sigmoid, gate), plus the
// float32 maxima of the residual compute_best add (:1147-1152).
//
// Kernel: k_cgate_p (phase-split, see the comment block in front of it); redo / multi-rank helpers.
#pragma once
#include "proj_p.hpp"

namespace s5 {

using v2i16 = __attribute__((ext_vector_type(2))) short;

struct CGateArgs {
    const int16_t *u;    // (N,H) SSM input (written by the B projection)
    BnArgs bn;           // GBN instantiations: this layer's BatchNorm; u is recomputed from `skip` and `u` is not read
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
    int32_t bad_bits; // status bits raised when a state is out of range (k_cgate_p)
    int32_t live_slots; // S16, > 0: state slots at or above it are zero and not in the stream (scan_quad.hpp ScanPairLArgs)
    int32_t t_lo, t_len; // k_cgate_p: the step range this launch covers (StepRange)
    const int32_t *sigtab; // [2][7 << sig_x]: gate operand r for a non-positive / positive sigmoid input (k_cgate_p)
    const int32_t *run_if; // WIDE (exact re-run): do the work only when *run_if != 0 (nullptr: always)
    int32_t mx_slot;       // first of the three LayerDyn::mx slots that receive the maxima (8; 11 for the re-run)
    const int16_t *sigdir; // DIRECT: [1 << sigdir_bits] gate operand r for every value of (gq >> (out_exp - sig_x))
    int32_t sigdir_bits;
};

// multi-rank mode only: the residual maxima of a re-run layer live in slots 11..13; move them to 8..10, the
// slots the ranks exchange
__global__ void k_select_maxima(LayerDyn *d)
{
    if (threadIdx.x < 3 && blockIdx.x == 0 && d->redo) d->mx[8 + threadIdx.x] = d->mx[11 + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Phase-split version (see proj_p.hpp for the idea).  Workgroup = one wave per (32-frame half, 32-channel tile): six
// waves at H=96, twelve at H=192; tile = 64 frames:
//   A   all threads: one 32-byte stream item (state p, 4 steps, re+im) per thread -> range check, complex
//       ReLU, 4x4 transpose inside the lane quad (DPP) so that a lane holds 4 consecutive states of ONE
//       frame -> byte planes S[frame][re P | im P] in LDS;
//   B1  wave (half, ct): C projection of its 32 channels x 32 frames, weights in registers, epilogue ->
//       x1 (kept in registers for the gate) and its byte planes X1[frame][H] in LDS;
//   B2  the same wave: out2 for the same channels/frames from the X1 planes (natural k order), LUT sigmoid,
//       gate, int16 store, residual maxima.
// No weights in LDS (35 KB at dim 0.5, 65 KB at dim 1.0), ~128 registers.
// LDS: [cs_re][cs_im][D][cs_out2][bias_eff] (Np ints each) [lut pairs 8] [S hi][S lo][X1 hi][X1 lo] [red 3x16]
// ---------------------------------------------------------------------------------------------
// The barriers between the phases order LDS traffic only (planes written by some waves, read by others).  __syncthreads()
// is a fence + barrier and the fence waits for EVERY outstanding memory operation (s_waitcnt vmcnt(0)): the loads requested
// a tile ahead would be drained at the next barrier.  This one waits for the wave's LDS operations and nothing else.
__device__ __forceinline__ void lds_barrier()
{
#ifdef S5_BARRIER_VM0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int CTRL>
__device__ __forceinline__ int32_t quad_xchg(int32_t v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// 4x4 transpose of w[0..3] across the four lanes of a quad: lane q's w[j] <-> lane j's w[q]
__device__ __forceinline__ void quad_transpose(int32_t (&w)[4], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    { // bit 0: register pairs (0,1), (2,3); partner = lane ^ 1
        const int32_t r01 = quad_xchg<0xB1>(b0 ? w[0] : w[1]), r23 = quad_xchg<0xB1>(b0 ? w[2] : w[3]);
        w[0] = b0 ? r01 : w[0]; w[1] = b0 ? w[1] : r01;
        w[2] = b0 ? r23 : w[2]; w[3] = b0 ? w[3] : r23;
    }
    { // bit 1: register pairs (0,2), (1,3); partner = lane ^ 2
        const int32_t r02 = quad_xchg<0x4E>(b1 ? w[0] : w[2]), r13 = quad_xchg<0x4E>(b1 ? w[1] : w[3]);
        w[0] = b1 ? r02 : w[0]; w[2] = b1 ? w[2] : r02;
        w[1] = b1 ? r13 : w[1]; w[3] = b1 ? w[3] : r13;
    }
}

// ---- packed 16-bit helpers of the PK16 epilogues (semantics probed on MI355X: tools/probe_pk16.hip -- the clamped forms
// saturate the EXACT result, v_cvt_pk_i16_i32 saturates each half, the SDWA forms sign-extend the selected half)
__device__ __forceinline__ uint32_t pk_cvt(int32_t lo, int32_t hi) // (sat16(lo), sat16(hi))
{
    uint32_t r;
    asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad_sat(uint32_t a, uint32_t m, uint32_t c) // sat16(a * m + c) per half
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(m), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_ashr(uint32_t a, uint32_t s) // s = shift in both halves
{
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(s), "v"(a));
    return r;
}
// the same with a wave-uniform second operand taken from an SGPR (no v_mov per use)
__device__ __forceinline__ uint32_t pk_mul_sat_u(uint32_t a, uint32_t m_uniform) // sat16(a * m) per half
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, 0 clamp" : "=v"(r) : "v"(a), "s"(m_uniform));
    return r;
}
__device__ __forceinline__ uint32_t pk_ashr_u(uint32_t a, uint32_t s_uniform)
{
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "s"(s_uniform), "v"(a));
    return r;
}
template <int HALF>
__device__ __forceinline__ int32_t mul24_h(int32_t a, uint32_t pk) // a * sext(half HALF of pk)
{
    int32_t r;
    if (HALF == 0)
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(pk));
    else
        asm("v_mul_i32_i24_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(pk));
    return r;
}
template <int HALF>
__device__ __forceinline__ float cvtf_h(uint32_t pk) // float(sext(half))
{
    float r;
    if (HALF == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(pk));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(pk));
    return r;
}
template <int HALF>
__device__ __forceinline__ int32_t ashr_h(int32_t s, uint32_t pk) // sext(half) >> s, s wave-uniform (an SGPR operand)
{
    int32_t r;
    if (HALF == 0)
        asm("v_ashrrev_i32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "s"(s), "v"(pk));
    else
        asm("v_ashrrev_i32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "s"(s), "v"(pk));
    return r;
}

constexpr int SIGTAB_WORDS = 2 * 7 * 64; // sig_x <= 6 on this path (host-checked)
constexpr int SIGDIR_MAX_BITS = 12, SIGDIR_BYTES = 2 << SIGDIR_MAX_BITS;
__host__ __device__ __forceinline__ int sigdir_lds_bytes(int bits) { return ((2 << bits) + 15) & ~15; }

// S16: the state stream holds int16, written with saturation by k_scan_quad_asm16 (a.xmax <= 32766 then: a saturated
// state fails the range check like any other state beyond the bound)
// DIRECT: the sigmoid input xx = gq >> (out_exp - sig_x) has only out_bits - (out_exp - sig_x) <= 12 bits, so r is read
// from a table over xx itself (no |xx|, segment index, remainder or sign logic at all)
// FTP: frames per tile, 64 (two 32-frame halves, 2*NT waves) or 32 (NT waves: smaller workgroups, more of them per CU)
// WIDE: the exact variant for states of any width (the re-run behind the range check): int32 states are split into
// FOUR byte planes, a = b3*2^24 + (b2'+128)*2^16 + (b1'+128)*2^8 + (b0'+128) with b3 signed and b' = byte ^ 0x80, so
// sum a*w = Horner over four MFMA passes with the same per-channel constant 128*sum(w) added at each of the three shifts
// -- all modulo 2^32, which is the reference's int32 matmul (fxparray.py:662).  No range check, complex ReLU through
// float32 exactly as the reference does it (fxp_prims.hpp crelu).
// PAIR (with S16): the states come from k_scan_pair_asm in pair-native order (scan_quad.hpp)
// PK16 (with DIRECT): y, out2 output, the gate's l operand and its result are all 16 bit, no out2 input conversion and
// every |bias_eff| fits 16 bits (host-checked, s5fxp_fast.hpp): both epilogues run on packed int16 pairs -- saturating
// packs, clamped packed sub / mad / add, SDWA half-word operands -- about a third fewer VALU instructions, same results
// GBN (with PK16): the SSM input u = BatchNorm(layer input) is recomputed here from `skip` -- the layer input this kernel reads
// anyway for the residual maxima -- with the exponents the B projection left in LayerDyn, instead of being written by the B
// projection and read back: -2 x N x H x 2 bytes of traffic per layer for ~12 more VALU operations per element (mfma_bn.hpp bn16_x4)
template <int KS, int NT, bool TRACE, bool S16 = false, bool DIRECT = false, int FTP = 64, bool WIDE = false, bool PAIR = false, bool PK16 = false,
          bool GBN = false>
// <= 128 registers: two six-wave workgroups per CU (at 136 only one was ever resident: measured)
// -DS5_CGATE_HID=1: states, u and skip of the NEXT tile all requested a tile ahead behind the compiler's back and waited for
// by exact count (scan_quad.hpp vm_wait).  Measured (profiles/r03_gate_prefetch_ab.txt): 226 us per 8-batch launch against
// 210 us for the default below, which requests the states at the top of the tile that uses them and lets the compiler's
// vmcnt(0) there drain the skip prefetch as well -- MORE bytes in flight make this kernel slower, not faster (the same
// build with every barrier draining all loads: 245 us).  Kept as the experiment's record.
#ifndef S5_CGATE_HID
#define S5_CGATE_HID 0
#endif
// -DS5_CGATE_COAL=0: u, skip and z move between registers and memory in the MFMA accumulator's layout (lane = frame: every
// load / store instruction touches 64 rows with 8 bytes each) instead of through the LDS tiles described at COAL below
#ifndef S5_CGATE_COAL
#define S5_CGATE_COAL 1
#endif
#ifndef S5_CGATE_LB
#define S5_CGATE_LB 4
#endif
__global__ __launch_bounds__(FTP * 2 * NT, NT <= 3 && KS == 1 && !WIDE ? S5_CGATE_LB : 3) void k_cgate_p(const CGateArgs a_k, GroupOff go)
{
    CGateArgs a = a_k; // (the LUT is indexed by thread below: that read stays on the kernel argument, so that this copy lives in registers)
    {
        const int64_t g = blockIdx.y;
        gshift(a.u, g * go.ws); gshift(a.skip, g * go.ws); gshift(a.xs, g * go.ws); gshift(a.z, g * go.ws);
        gshift(a.skip_e.dyn, g * go.ws); gshift(a.dynw, g * go.ws); gshift(a.run_if, g * go.ws); gshift(a.status, g * go.status);
        if constexpr (GBN) { gshift(a.bn.dyn, g * go.ws); gshift(a.bn.xe.dyn, g * go.ws); }
    }
    static_assert(!GBN || (PK16 && !TRACE && !WIDE), "the BatchNorm rides on the packed-epilogue kernel only");
    constexpr int P = 32 * KS, H = 32 * NT, FT = FTP, NW = (FT / 32) * NT, NTHR = 64 * NW; // one wave per (half, column tile)
    constexpr int KPS = 2 * P + 16, KPX = H + 16;
    constexpr int NU = 1, SUBSTEP = 0;   // units per wave
    constexpr int ITEMS = (FT / 4) * P, ROUNDS = (ITEMS + NTHR - 1) / NTHR;
    // hidden prefetches, see below (the dim 1.0 kernel on all 128 state slots has no registers left for them)
    constexpr bool HID = S5_CGATE_HID && S16 && PAIR && PK16 && !WIDE && !TRACE && KS * NT < 24;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int32_t *csr = reinterpret_cast<int32_t *>(smem), *csi = csr + H, *Dl = csi + H, *cs2 = Dl + H, *be = cs2 + H, *lutp = be + H;
    int32_t *sigt = lutp + 8; // SIGTAB_WORDS, or the direct table (int16, SIGDIR_BYTES)
    const int16_t *sigd = reinterpret_cast<const int16_t *>(sigt);
    constexpr int NPL = WIDE ? 4 : 2; // byte planes of the state operand; plane NPL-1 is the signed top byte
    // the direct table takes what it needs (2 bytes x 2^sigdir_bits, to a multiple of 16), not the 8 KB of its widest form
    int8_t *Sbase = reinterpret_cast<int8_t *>(sigt) + (DIRECT ? sigdir_lds_bytes(a.sigdir_bits) : 4 * SIGTAB_WORDS);
    int8_t *Sl = Sbase, *Sh = Sbase + (NPL - 1) * FT * KPS, *Xh = Sbase + NPL * FT * KPS, *Xl = Xh + FT * KPX;
    float *red = reinterpret_cast<float *>(Xl + FT * KPX);
    int32_t *bntab = reinterpret_cast<int32_t *>(red + 48); // GBN: 4 * H BatchNorm operands (bn16_setup)
    // COAL: u and skip come in, and z goes out, as whole rows -- 16 bytes per lane, a wave instruction covers 1 KB of
    // consecutive addresses -- through two LDS tiles [frame][TROW]; the epilogues, whose lanes are FRAMES (the accumulator
    // layout of the channel-major MFMA), pick their 8-byte (frame, four channels) pieces out of LDS.  In the accumulator's own
    // layout every global load / store instruction touched 64 different rows with 8 bytes each: the bytes per batch are the
    // same, the memory pipeline sees an eighth of the requests.  TROW: 8-byte reads by 32 lanes 200 bytes apart hit 32 bank pairs.
    constexpr bool COAL = S5_CGATE_COAL && PK16 && !TRACE && !WIDE && !GBN && !HID;
    constexpr int TROW = 2 * H + 8, VPF = H / 8, NVC = FT * VPF / NTHR; // bytes per tile row; 16-byte vectors per frame / per thread
    static_assert(!COAL || FT * VPF % NTHR == 0, "tile vectors per thread");
    int8_t *Ut = reinterpret_cast<int8_t *>(red + 48), *St = Ut + FT * TROW; // u (then z) and skip of the current tile
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int ct = wave % NT, sub0 = wave / NT;
    const StepRange sr{a.t_lo, a.t_len};
    const int64_t tiles = (a.N / a.L) * ((sr.t_len + FT - 1) / FT);
    if (WIDE && a.run_if && *a.run_if == 0) return;

    // weights of this wave's 32 channels (A operand rows), all k-steps, in registers
    v4i wre[KS], wim[KS], wo2[NT];
    {
        const size_t row = (size_t)(32 * ct + r);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wre[ks] = *reinterpret_cast<const v4i *>(a.w_re.wt + row * a.w_re.Kp + 32 * ks + 16 * h);
            wim[ks] = *reinterpret_cast<const v4i *>(a.w_im.wt + row * a.w_im.Kp + 32 * ks + 16 * h);
        }
#pragma unroll
        for (int ks = 0; ks < NT; ++ks) wo2[ks] = *reinterpret_cast<const v4i *>(a.w_o2.wt + row * a.w_o2.Kp + 32 * ks + 16 * h);
    }
    for (int i = threadIdx.x; i < H; i += NTHR) {
        csr[i] = a.w_re.cs128[i]; csi[i] = a.w_im.cs128[i]; Dl[i] = a.D[i]; cs2[i] = a.w_o2.cs128[i];
        if (!PK16) be[i] = a.bias_eff[i];
    }
    if (PK16) // packed pairs (every value fits 16 bits: host-checked)
        for (int i = threadIdx.x; i < H / 2; i += NTHR)
            be[i] = (int32_t)(((uint32_t)a.bias_eff[2 * i] & 0xffffu) | ((uint32_t)a.bias_eff[2 * i + 1] << 16));
    if (threadIdx.x < 8) lutp[threadIdx.x] = a_k.lut[threadIdx.x] | (a_k.lut[threadIdx.x < 7 ? threadIdx.x + 1 : 7] << 16);
    if (DIRECT) {
        for (int i = threadIdx.x; i < (1 << a.sigdir_bits) / 2; i += NTHR)
            sigt[i] = reinterpret_cast<const int32_t *>(a.sigdir)[i];
    } else {
        for (int i = threadIdx.x; i < (14 << a.sig_x); i += NTHR) sigt[i] = a.sigtab[i];
    }
    Bn16 bn{};
    if constexpr (GBN) bn = bn16_setup(a.bn, *a.bn.dyn, bntab, H); // the B projection of this layer has published the exponents
    const int dsh = a.out_exp - a.sig_x, dbias = 1 << (a.sigdir_bits - 1);
    const int skip_e = a.skip_e.get();
    const float kz = ldexpf(1.f, skip_e - a.res_exp); // fz + fs = 2^-skip_e * (z * kz + s), exactly
    const int sx = a.sig_x, S = 1 << sx;
    // out2 input conversion (fxpmodel.py:335-347) as uniform shift/clip operands; identity when not needed
    const int cv_l = a.conv && a.inp_exp > a.y_exp ? a.inp_exp - a.y_exp : 0, cv_r = a.conv && a.y_exp > a.inp_exp ? a.y_exp - a.inp_exp : 0;
    const int cv_b1 = a.conv && a.inp_exp != a.y_exp ? a.y_bits : 32, cv_b2 = a.conv && a.y_bits > a.inp_bits ? a.inp_bits : 32;
    const int cv_b = cv_b1 < cv_b2 ? cv_b1 : cv_b2;
    // PK16: change_cfg(x1 -> l operand) with equal widths is a left shift that saturates or a right shift (fxp_prims.hpp chcfg)
    const int lq_l = a.l_exp > a.y_exp ? a.l_exp - a.y_exp : 0, lq_r = a.y_exp > a.l_exp ? a.y_exp - a.l_exp : 0;
    uint32_t xrange = 0;
    v2i16 pmax = {0, 0}, pmin = {0, 0}; // S16: running extremes of (re, im) as packed int16
    float mx[3] = {0.f, 0.f, 0.f}; // [0]: |z*kz + s| as converted integers, scaled once at the end; [1], [2] stay 0
    const int ch0 = 32 * ct + 4 * h;
    // u and skip of this wave's unit (32 frames x 32 channels), 8 bytes per lane and 8-channel group.  They are requested a
    // whole tile ahead and IN TURN: the next tile's u goes into the registers the first epilogue has just emptied, the next
    // tile's skip into those the second epilogue has emptied -- no extra registers, and the kernel has loads in flight during
    // its arithmetic phases instead of one burst at the top of every tile (32 KB per tile and workgroup outstanding for a third
    // of the tile's time is all that two workgroups per CU had in flight: ~3 TB/s by Little's law, which is what it ran at)
    v2i uq[NU][4], sq[NU][4];
    v4i urow[COAL ? NVC : 1], srow[COAL ? NVC : 1]; // COAL: this thread's vectors of the NEXT tile's u and skip rows
    auto load_tile = [&](v4i(&dst)[COAL ? NVC : 1], const int16_t *src, const TileWalk<FT> &tw) {
        const int64_t b = tw.b;
        const int t = tw.t(sr), nv = tw.nvalid(sr);
        const char *base = reinterpret_cast<const char *>(src + (b * a.L + t) * H); // wave-uniform
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int v = threadIdx.x + NTHR * i;
            int f = v / VPF;
            f = f < nv ? f : nv - 1;
            dst[i] = *reinterpret_cast<const v4i *>(base + 2u * (unsigned)(f * H + 8 * (v % VPF)));
        }
    };
    auto load_rows = [&](v2i(&dst)[NU][4], const int16_t *src, const TileWalk<FT> &tw) {
        const int64_t b = tw.b;
        const int t = tw.t(sr), nv = tw.nvalid(sr);
        const char *base = reinterpret_cast<const char *>(src + (b * a.L + t) * H); // wave-uniform
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int fo = 32 * (sub0 + u * SUBSTEP) + r;
            fo = fo < nv ? fo : nv - 1;
            const unsigned fb = 2u * (unsigned)(fo * H + ch0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (HID) dst[u][g] = gload8_hidden(base, fb + 16 * g);
                else dst[u][g] = *reinterpret_cast<const v2i *>(base + fb + 16 * g);
            }
        }
    };
    // ... and so are the recurrence's states on the pair rung (the shipped path): phase A's two 8-byte loads per item used to
    // be issued and consumed on the spot, every tile opening with one exposed round trip to memory
    // On that path the three prefetches are issued behind the compiler's back and waited for by count (scan_quad.hpp
    // vm_wait): the wait its own pass puts in front of their first use, a tile later and behind conditional stores, is
    // vmcnt(0) -- which also drains whatever was requested since.  Memory operations of a wave on a full tile, in order:
    //   phase A: [x(next): NX]   B1: [u(next): 4]   B2: [z stores: 4] [skip(next): 4]
    constexpr bool XPRE = HID;
    constexpr int NX_MIN = 2 * (ITEMS / NTHR); // x loads every wave issues per tile (waves of the last round: two more)
    v2i xq[XPRE ? ROUNDS : 1][2];
    auto load_x = [&](const TileWalk<FT> &tw) {
        const int64_t b = tw.b;
        const int t = tw.t(sr), nv = tw.nvalid(sr);
        const char *xb = reinterpret_cast<const char *>(reinterpret_cast<const int16_t *>(a.xs) + (pair_word(b, t >> 3, 0, a.TB >> 1, P) << 1));
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int q = threadIdx.x + NTHR * i;
            if (ROUNDS * NTHR == ITEMS || q < ITEMS) {
                const int grp = q / P, p = q % P;
                int o = 4 * grp;
                if (o >= nv) o = (nv - 1) & ~3;
                const unsigned xo = 2u * (unsigned)((((((p >> 5) * (a.TB >> 1) + (o >> 3)) << 5) + (p & 31)) << 4) + (o & 4));
                xq[i][0] = xq[i][1] = v2i{0, 0};
                if (a.live_slots <= 0 || p < a.live_slots) {
                    xq[i][0] = gload8_hidden(xb, xo);
                    xq[i][1] = gload8_hidden(xb, xo + 16);
                }
            }
        }
    };
    TileWalk<FT> walk((int64_t)blockIdx.x, sr, gridDim.x);
    if ((int64_t)blockIdx.x < tiles) {
        if constexpr (XPRE) load_x(walk);
        if constexpr (COAL) {
            load_tile(urow, a.u, walk);
            load_tile(srow, a.skip, walk);
        } else {
            if constexpr (!GBN) load_rows(uq, a.u, walk);
            load_rows(sq, a.skip, walk);
        }
    }
    __syncthreads();
    char *zb_prev = nullptr; // COAL: where the z tile still sitting in LDS belongs (the previous tile of this workgroup)
    int nvalid_prev = 0;
    // COAL: a thread moves the SAME vectors of the z tile out and of the u tile in, so the tile changes owner without a barrier
    auto tiles_in_out = [&](bool incoming) {
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int v = threadIdx.x + NTHR * i, f = v / VPF, og = v % VPF;
            int8_t *cu = Ut + f * TROW + 16 * og, *cs_ = St + f * TROW + 16 * og;
            if (zb_prev) {
                const v2i z0 = *reinterpret_cast<const v2i *>(cu), z1 = *reinterpret_cast<const v2i *>(cu + 8);
                if (f < nvalid_prev) *reinterpret_cast<v4i *>(zb_prev + 2u * (unsigned)(f * H + 8 * og)) = v4i{z0[0], z0[1], z1[0], z1[1]};
            }
            if (incoming) {
                *reinterpret_cast<v2i *>(cu) = v2i{urow[i][0], urow[i][1]};
                *reinterpret_cast<v2i *>(cu + 8) = v2i{urow[i][2], urow[i][3]};
                *reinterpret_cast<v2i *>(cs_) = v2i{srow[i][0], srow[i][1]};
                *reinterpret_cast<v2i *>(cs_ + 8) = v2i{srow[i][2], srow[i][3]};
            }
        }
    };

    prologue_loads_done();
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x, walk.advance()) {
        const int64_t b0 = walk.b;
        const int t0 = walk.t(sr), nvalid = walk.nvalid(sr);
        const TileWalk<FT> walk_next = walk.next();
        const int64_t n0 = b0 * a.L + t0;
        // wave-uniform base of this tile; everything below is a 32-bit byte offset from it (no 64-bit address arithmetic per
        // thread: the kernel sits at its register cap)
        char *zb = reinterpret_cast<char *>(a.z + n0 * H);
        const int64_t tile_next = tile + gridDim.x;
        if constexpr (HID) vm_wait<12>(xq); // newer than this tile's states: u, the last tile's stores, skip
        if constexpr (COAL) tiles_in_out(true); // z of the previous tile out, u and skip of this one in
        // ---- phase A: stream items -> byte planes
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int q = threadIdx.x + NTHR * i;
            if (ROUNDS * NTHR == ITEMS || q < ITEMS) {
                const int grp = q / P, p = q % P;
                int o = 4 * grp;
                if (o >= nvalid) o = (nvalid - 1) & ~3; // partial tile: re-read the last block (results unused)
                // steps of the sequence's last block beyond its length (L % 4 != 0): the recurrence ran on through them
                // from whatever the stream holds there; they must not reach the range check
                const int nlive = nvalid - o; // >= 1; >= 4 everywhere but in that block
                int32_t w[4];
                if (WIDE) {
                    const int32_t *src = a.xs + native_word(b0, t0 + o, p, 0, a.TB, P);
                    const v4i cre = *reinterpret_cast<const v4i *>(src), cim = *reinterpret_cast<const v4i *>(src + 4);
                    int32_t xr[4] = {cre[0], cre[1], cre[2], cre[3]}, xi[4] = {cim[0], cim[1], cim[2], cim[3]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) crelu(xr[j], xi[j]);
                    quad_transpose(xr, l);
                    quad_transpose(xi, l);
                    // this lane = frame 4*grp + (l&3); xr[m], xi[m] = state (p & ~3) + m: four byte planes each
                    const int row = (4 * grp + (l & 3)) * KPS + (p & ~3);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int32_t(&v)[4] = c ? xi : xr;
                        const unsigned t01 = perm((unsigned)v[1], (unsigned)v[0], 0x05010400u), u01 = perm((unsigned)v[1], (unsigned)v[0], 0x07030602u);
                        const unsigned t23 = perm((unsigned)v[3], (unsigned)v[2], 0x05010400u), u23 = perm((unsigned)v[3], (unsigned)v[2], 0x07030602u);
                        int8_t *dst = Sbase + row + c * P;
                        *reinterpret_cast<int32_t *>(dst) = (int32_t)(perm(t23, t01, 0x05040100u) ^ 0x80808080u);
                        *reinterpret_cast<int32_t *>(dst + FT * KPS) = (int32_t)(perm(t23, t01, 0x07060302u) ^ 0x80808080u);
                        *reinterpret_cast<int32_t *>(dst + 2 * FT * KPS) = (int32_t)(perm(u23, u01, 0x05040100u) ^ 0x80808080u);
                        *reinterpret_cast<int32_t *>(dst + 3 * FT * KPS) = (int32_t)perm(u23, u01, 0x07060302u);
                    }
                    continue;
                }
                if (S16) {
                    if (PAIR) {
                        // one 8-step item per lane of the pair: this thread takes the half with its 4 steps from both.
                        // lane A: [im0 im2 | re1 re3], lane B: [re0 re2 | im1 im3] (per half)
                        // the tile's items of state group p >> 5 start at pair_word(b0, t0 >> 3, 32 (p >> 5), ...): uniform base + 32-bit offset
                        v2i qa, qb;
                        if constexpr (XPRE) {
                            qa = xq[i][0]; qb = xq[i][1]; // load_x: requested a tile ago
                        } else {
                            const char *xb = reinterpret_cast<const char *>(reinterpret_cast<const int16_t *>(a.xs) + (pair_word(b0, t0 >> 3, 0, a.TB >> 1, P) << 1));
                            const unsigned xo = 2u * (unsigned)((((((p >> 5) * (a.TB >> 1) + (o >> 3)) << 5) + (p & 31)) << 4) + (o & 4));
                            qa = qb = v2i{0, 0};
                            if (a.live_slots <= 0 || p < a.live_slots) {
                                qa = *reinterpret_cast<const v2i *>(xb + xo); qb = *reinterpret_cast<const v2i *>(xb + xo + 16);
                            }
                        }
                        w[0] = (int32_t)perm((unsigned)qa[0], (unsigned)qb[0], 0x05040100u);
                        w[1] = (int32_t)perm((unsigned)qb[1], (unsigned)qa[1], 0x05040100u);
                        w[2] = (int32_t)perm((unsigned)qa[0], (unsigned)qb[0], 0x07060302u);
                        w[3] = (int32_t)perm((unsigned)qb[1], (unsigned)qa[1], 0x07060302u);
                    } else {
                    // 16 bytes: re of steps 0..3, then im of steps 0..3, as int16; w[j] = re_j | im_j << 16 by two perms
                    v4i q4 = {0, 0, 0, 0};
                    if (a.live_slots <= 0 || p < a.live_slots)
                        q4 = *reinterpret_cast<const v4i *>(reinterpret_cast<const int16_t *>(a.xs) + native_word(b0, t0 + o, p, 0, a.TB, P));
                    w[0] = (int32_t)perm((unsigned)q4[2], (unsigned)q4[0], 0x05040100u);
                    w[1] = (int32_t)perm((unsigned)q4[2], (unsigned)q4[0], 0x07060302u);
                    w[2] = (int32_t)perm((unsigned)q4[3], (unsigned)q4[1], 0x05040100u);
                    w[3] = (int32_t)perm((unsigned)q4[3], (unsigned)q4[1], 0x07060302u);
                    }
                    if (nlive < 4) { // tile-uniform except in a sequence's last tile
#pragma unroll
                        for (int j = 1; j < 4; ++j) w[j] = j < nlive ? w[j] : 0;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pmax = __builtin_elementwise_max(pmax, __builtin_bit_cast(v2i16, w[j]));
                        pmin = __builtin_elementwise_min(pmin, __builtin_bit_cast(v2i16, w[j]));
                        // lexicographic (re, im) > (0, 0)  <=>  re * 2^16 + im > 0 (|im| < 2^15 cannot outweigh re != 0; the
                        // sum wraps only for re = -32768, a value beyond xmax: such a tile raises `redo` and is discarded)
                        const bool keep = ((int32_t)((uint32_t)w[j] << 16) + (w[j] >> 16)) > 0;
                        w[j] = keep ? w[j] : 0;
                    }
                } else {
                const int32_t *src = a.xs + native_word(b0, t0 + o, p, 0, a.TB, P);
                const v4i cre = *reinterpret_cast<const v4i *>(src), cim = *reinterpret_cast<const v4i *>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int32_t xr = j < nlive ? cre[j] : 0, xi = j < nlive ? cim[j] : 0;
                    const uint32_t ur = (uint32_t)(xr + a.xmax), ui = (uint32_t)(xi + a.xmax);
                    xrange = xrange > ur ? xrange : ur;
                    xrange = xrange > ui ? xrange : ui;
                    // complex ReLU = lexicographic max(z, 0); exact in integers while |x| <= xmax < 2^24
                    const bool keep = (xr > 0) | ((xr == 0) & (xi > 0));
                    w[j] = keep ? (int32_t)perm((unsigned)xi, (unsigned)xr, 0x05040100u) : 0;
                }
                }
                quad_transpose(w, l);
                // now: this lane = frame 4*grp + (l&3), w[m] = (re | im << 16) of state (p & ~3) + m
                const unsigned t01 = perm((unsigned)w[1], (unsigned)w[0], 0x05010400u), u01 = perm((unsigned)w[1], (unsigned)w[0], 0x07030602u);
                const unsigned t23 = perm((unsigned)w[3], (unsigned)w[2], 0x05010400u), u23 = perm((unsigned)w[3], (unsigned)w[2], 0x07030602u);
                const int row = (4 * grp + (l & 3)) * KPS + (p & ~3);
                *reinterpret_cast<int32_t *>(Sl + row) = (int32_t)(perm(t23, t01, 0x05040100u) ^ 0x80808080u);
                *reinterpret_cast<int32_t *>(Sh + row) = (int32_t)perm(t23, t01, 0x07060302u);
                *reinterpret_cast<int32_t *>(Sl + row + P) = (int32_t)(perm(u23, u01, 0x05040100u) ^ 0x80808080u);
                *reinterpret_cast<int32_t *>(Sh + row + P) = (int32_t)perm(u23, u01, 0x07060302u);
            }
        }
        if constexpr (XPRE) {
            if (tile_next < tiles) load_x(walk_next);
        }
        if constexpr (COAL) {
            if (tile_next < tiles) { // the registers are free again: the next tile's rows, a whole tile ahead
                load_tile(urow, a.u, walk_next);
                load_tile(srow, a.skip, walk_next);
            }
        }
        lds_barrier();
        if constexpr (HID) vm_wait<8>(uq); // newer: the last tile's stores, skip (and x(next), if there is a next tile)
        // ---- phase B1: C projection + first epilogue
        int32_t x1v[NU][16];
        uint32_t x1p[NU][8]; // PK16: the same values as int16 pairs (channels 2q, 2q+1 of group g at [2g + q])
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int sub = sub0 + u * SUBSTEP;
            const int64_t n = n0 + 32 * sub + r;
            const int8_t *row0 = Sbase + (32 * sub + r) * KPS + 16 * h;
            v16i are, aim;
            mfma_nplanes<KS, NPL>(are, wre, row0, FT * KPS, csr + ch0);
            mfma_nplanes<KS, NPL>(aim, wim, row0 + P, FT * KPS, csi + ch0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i Dv = *reinterpret_cast<const v4i *>(Dl + ch0 + 8 * g);
                const int off = (32 * sub + r) * KPX + ch0 + 8 * g;
                if constexpr (PK16) {
                    // fxpmodel.py:746-793 + :1125 on int16 pairs: cx = sat(sat(cr) - sat(ci)); y = sat(2 cx + sat(D u)); x1 = max(y, 0)
                    int32_t ubn[4];
                    if constexpr (GBN) { // u = BatchNorm(layer input) of these four channels, as the B projection computes it
                        int32_t hin[4], tbn[4];
                        unpack4_i16(sq[u][g], hin);
                        bn16_x4(bn, hin, ch0 + 8 * g, tbn, ubn);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t crp = pk_cvt(asr(are[4 * g + 2 * q], a.rs_re), asr(are[4 * g + 2 * q + 1], a.rs_re));
                        const uint32_t cip = pk_cvt(asr(aim[4 * g + 2 * q], a.rs_im), asr(aim[4 * g + 2 * q + 1], a.rs_im));
                        uint32_t dup;
                        if constexpr (GBN) {
                            dup = pk_cvt(asr(__mul24(Dv[2 * q], ubn[2 * q]), a.rs_d), asr(__mul24(Dv[2 * q + 1], ubn[2 * q + 1]), a.rs_d));
                        } else {
                            uint32_t upk;
                            if constexpr (COAL) upk = (uint32_t)(*reinterpret_cast<const v2i *>(Ut + (32 * sub + r) * TROW + 2 * (ch0 + 8 * g)))[q];
                            else upk = (uint32_t)uq[u][g][q];
                            dup = pk_cvt(asr(mul24_h<0>(Dv[2 * q], upk), a.rs_d), asr(mul24_h<1>(Dv[2 * q + 1], upk), a.rs_d));
                        }
                        const uint32_t yp = pk_mad_sat(pk_sub_sat(crp, cip), 0x00020002u, dup); // 2*cx is not clipped, :765-767
                        x1p[u][2 * g + q] = pk_max(yp, 0u);
                    }
                    const uint32_t p01 = x1p[u][2 * g], p23 = x1p[u][2 * g + 1];
                    *reinterpret_cast<int32_t *>(Xl + off) = (int32_t)(perm(p23, p01, 0x06040200u) ^ 0x80808080u);
                    *reinterpret_cast<int32_t *>(Xh + off) = (int32_t)perm(p23, p01, 0x07050301u);
                } else {
                int32_t uv[4], xv[4];
                unpack4_i16(uq[u][g], uv);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int32_t cr = sat(asr(are[4 * g + e], a.rs_re), a.y_bits);
                    const int32_t ci = sat(asr(aim[4 * g + e], a.rs_im), a.y_bits);
                    const int32_t cx = sat(cr - ci, a.y_bits);
                    const int32_t du = sat(asr(__mul24(Dv[e], uv[e]), a.rs_d), a.y_bits);
                    const int32_t y = sat(2 * cx + du, a.y_bits); // 2*cx is not clipped, fxpmodel.py:765-767
                    if (TRACE) {
                        if (a.tr_ys && 32 * sub + r < nvalid) a.tr_ys[n * H + ch0 + 8 * g + e] = y;
                    }
                    const int32_t x1 = y < 0 ? 0 : y;
                    x1v[u][4 * g + e] = x1;
                    xv[e] = x1;
                }
                if (a.conv) { // uniform: the out2 input conversion is usually the identity
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = sat(asr(wshl(xv[e], cv_l), cv_r), cv_b);
                }
                const unsigned p01 = perm((unsigned)xv[1], (unsigned)xv[0], 0x05010400u), p23 = perm((unsigned)xv[3], (unsigned)xv[2], 0x05010400u);
                *reinterpret_cast<int32_t *>(Xl + off) = (int32_t)(perm(p23, p01, 0x05040100u) ^ 0x80808080u);
                *reinterpret_cast<int32_t *>(Xh + off) = (int32_t)perm(p23, p01, 0x07060302u);
                }
            }
        }
        if constexpr (!GBN) {
            if (!COAL && tile_next < tiles) load_rows(uq, a.u, walk_next); // the first epilogue is done with u
        }
        lds_barrier();
        if constexpr (HID) { // newer: x(next) and u(next), if there is a next tile
            if (tile_next < tiles) vm_wait<NX_MIN + 4>(sq);
            else vm_wait<0>(sq);
        }
        // ---- phase B2: out2 + second epilogue
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int sub = sub0 + u * SUBSTEP;
            const int64_t n = n0 + 32 * sub + r;
            v16i acc;
#ifdef S5_GATE_CHECK
            bool live2 = false; // out2's operand: one decision for the whole row of fragments (mfma_planes takes them all)
#pragma unroll
            for (int ks = 0; ks < NT; ++ks) live2 |= gate_check<2>(Xl + (32 * sub + r) * KPX + 16 * h + 32 * ks, -(FT * KPX), 2);
            if (!live2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0;
            } else
#endif
            mfma_planes<NT>(acc, wo2, Xh + (32 * sub + r) * KPX + 16 * h, Xl + (32 * sub + r) * KPX + 16 * h, cs2 + ch0);
            auto b2_pk16 = [&]() {
                // out2 bias, table sigmoid, gate (fxpmodel.py:1133-1137, :97-144, :1075-1093) on int16 pairs
                const uint32_t lm = 0x10001u * (uint32_t)(1 << lq_l), lr = 0x10001u * (uint32_t)lq_r;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = ch0 + 8 * g;
                    const v2i bp = *reinterpret_cast<const v2i *>(be + (ch >> 1));
                    v2i zo;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        uint32_t gp = pk_cvt(asr(acc[4 * g + 2 * q], a.rs_o2), asr(acc[4 * g + 2 * q + 1], a.rs_o2));
                        gp = pk_add_sat(gp, (uint32_t)bp[q]);
                        const int32_t r0 = sigd[ashr_h<0>(dsh, gp) + dbias], r1 = sigd[ashr_h<1>(dsh, gp) + dbias];
                        uint32_t lp = x1p[u][2 * g + q];
                        // change_cfg of the gate's l operand: a saturating left shift (x lm) or a right shift, never both -- applied
                        // as both (x 1 and >> 0 are the identity) rather than behind two uniform branches per pair
                        lp = pk_ashr_u(pk_mul_sat_u(lp, lm), lr);
                        const uint32_t zp = pk_cvt(asr(mul24_h<0>(r0, lp), a.rs_gate), asr(mul24_h<1>(r1, lp), a.rs_gate));
                        zo[q] = (int)zp;
                        uint32_t sp;
                        if constexpr (COAL) sp = (uint32_t)(*reinterpret_cast<const v2i *>(St + (32 * sub + r) * TROW + 2 * ch))[q];
                        else sp = (uint32_t)sq[u][g][q];
                        mx[0] = fmaxf(mx[0], fabsf(__fmaf_rn(cvtf_h<0>(zp), kz, cvtf_h<0>(sp))));
                        mx[0] = fmaxf(mx[0], fabsf(__fmaf_rn(cvtf_h<1>(zp), kz, cvtf_h<1>(sp))));
                    }
                    if constexpr (COAL) *reinterpret_cast<v2i *>(Ut + (32 * sub + r) * TROW + 2 * ch) = zo; // u is done with: B1 is behind a barrier
                    else *reinterpret_cast<v2i *>(zb + 2u * (unsigned)((32 * sub + r) * H + ch)) = zo;
                }
            };
            if constexpr (PK16) {
                if (!HID) {
                    if (32 * sub + r < nvalid) b2_pk16();
                } else if (nvalid == FT) {
                    b2_pk16(); // a full tile: no control flow around its stores, their number is known
                } else {
                    if (32 * sub + r < nvalid) b2_pk16();
                    prologue_loads_done(); // behind conditional stores nothing is left in flight
                }
            }
            if (!PK16 && 32 * sub + r < nvalid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = ch0 + 8 * g;
                    const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                    int32_t sv[4], o[4];
                    unpack4_i16(sq[u][g], sv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int32_t gq = sat(asr(acc[4 * g + e], a.rs_o2), a.out_bits);
                        gq = sat(gq + bv[e], a.out_bits);
                        // LUT sigmoid (fxp_prims.hpp sigmoid_lut) + change_cfg to the gate's r operand.  Both are
                        // functions of the sign of xx and of (min(|xx| >> sx, 6), |xx| mod 2^sx) only: 2 x 7 x 2^sx
                        // values, tabulated by the host with the same formula (s5fxp_fast.hpp).  The TRACE
                        // instantiation computes s the long way (it has to write it out).
                        int32_t s = 0, rq;
                        if (DIRECT) {
                            rq = sigd[(gq >> dsh) + dbias];
                        } else {
                        const int32_t xx = chexp(gq, a.out_bits, a.out_exp, sx);
                        const int32_t ax = xx < 0 ? -xx : xx;
                        int32_t ind = ax >> sx;
                        ind = ind > 6 ? 6 : ind;
                        const int32_t mu = ax & (S - 1);
                        if (TRACE) {
                            const uint32_t pr = (uint32_t)lutp[ind];
                            const int32_t half = (__mul24(S - mu, (int32_t)(pr & 0xffffu)) >> sx) + (__mul24(mu, (int32_t)(pr >> 16)) >> sx);
                            s = (1 << (a.sig_y - 1)) + (xx > 0 ? half : -half);
                            rq = chcfg(s, a.out_bits, a.sig_y, a.r_bits, a.r_exp);
                        } else {
                            rq = sigt[((ind << sx) | mu) + (xx > 0 ? 7 * S : 0)];
                        }
                        }
                        const int32_t lq = chcfg(x1v[u][4 * g + e], a.y_bits, a.y_exp, a.l_bits, a.l_exp);
                        const int32_t z = sat(asr(__mul24(lq, rq), a.rs_gate), a.res_bits);
                        if (TRACE) {
                            if (a.tr_out2) a.tr_out2[n * H + ch + e] = gq;
                            if (a.tr_sig) a.tr_sig[n * H + ch + e] = s;
                            if (a.tr_z) a.tr_z[n * H + ch + e] = z;
                        }
                        o[e] = z;
                        const float cz = (float)z, cs = (float)sv[e];
                        // only max |z + skip| chooses the exponent (fxparray.py:421-425); the operands' own maxima
                        // (slots 9, 10) merely size the reference's intermediate bit width and are not needed
                        mx[0] = fmaxf(mx[0], fabsf(__fmaf_rn(cz, kz, cs)));
                    }
                    *reinterpret_cast<v2i *>(zb + 2u * (unsigned)((32 * sub + r) * H + ch)) = pack4_i16(o[0], o[1], o[2], o[3]);
                }
            }
        }
        if constexpr (COAL) {
            zb_prev = zb; nvalid_prev = nvalid;
            lds_barrier(); // every wave's z pieces are in the tile, every wave is done with the skip tile
        } else {
            if (tile_next < tiles) load_rows(sq, a.skip, walk_next); // the second epilogue is done with skip
        }
    }
    if constexpr (COAL) tiles_in_out(false); // the last tile's z
    // ---- range flag and the three maxima (scaled back: power-of-two factors, exact)
    if (S16) {
        const int hi = pmax[0] > pmax[1] ? pmax[0] : pmax[1], lo = pmin[0] < pmin[1] ? pmin[0] : pmin[1];
        if (hi > a.xmax || lo < -a.xmax) xrange = 0xffffffffu;
    }
    if (__any(xrange > 2u * (uint32_t)a.xmax) && l == 0) {
        atomicExch(&a.dynw->redo, 1);
        atomicOr(a.status, a.bad_bits);
    }
    mx[0] = ldexpf(mx[0], -skip_e);
    mx[1] = ldexpf(mx[1], -a.res_exp);
    mx[2] = ldexpf(mx[2], -skip_e);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float x = mx[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
        if (l == 0) red[i * 16 + wave] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float x = red[threadIdx.x * 16];
        for (int w = 1; w < NW; ++w) x = fmaxf(x, red[threadIdx.x * 16 + w]);
        atomicMax(a.dynw->mx + a.mx_slot + threadIdx.x, __float_as_uint(x));
    }
}

} // namespace s5
