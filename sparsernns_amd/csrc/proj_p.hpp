// proj_p.hpp -- phase-split projection kernels: element work spread over the whole workgroup, one 32-column
// tile of the matmul per wave.
//
// The per-wave kernels in mfma_proj.hpp / mfma_bn.hpp keep a whole frame tile (all k-steps of byte planes,
// all column tiles of accumulators) in one wave: 200-256 registers, two waves per SIMD, and every stall of
// the serial load -> chain -> MFMA -> epilogue sequence is exposed.  Here a workgroup of four waves owns a
// tile of 64 frames:
//   phase A  all threads: 16-byte coalesced loads of the (64,H) int16 tile, the BatchNorm chain on eight
//            channels per vector, u stored with 16-byte coalesced stores, byte planes into LDS [frame][k];
//   phase B  wave w: column tile w of the matmul for both 32-frame halves.  The MFMA runs as D = X * W
//            (A operand = byte planes from LDS, B operand = this wave's weight columns, held in registers for
//            the whole kernel), so a lane owns ONE output channel and 4-frame groups of it -- exactly one
//            16-byte item of the scan-native stream per (group) and per-channel constants are per-lane
//            registers.
// Planes are double buffered: one barrier per tile, phase A of tile i+1 overlaps phase B of tile i in other
// waves.  ~30 KB LDS and < 128 registers: four workgroups (16 waves) per CU.
#pragma once
// -DS5_PHASE_PROF (tools/prof_phases.py, never in the shipped library): every thread 0 accumulates the shader clock between
// phase marks of k_enc_p.  A mark first touches a register the preceding work produced (a v_cmp into vcc: the hardware interlock makes
// it wait for an MFMA or a load that is still in flight -- a bare s_memtime is hoisted over pure arithmetic by the compiler
// and overtakes pending MFMAs in the hardware), then reads the clock.
#ifdef S5_PHASE_PROF
__device__ long long g_phase_prof[2048 * 8];
// all state in SGPRs (the kernels are at their register cap: per-lane accumulators would spill and the reloads would queue
// behind the prefetch loads); 32-bit deltas are enough for one launch
#define PHASE_DECL unsigned prof_last, prof_a0 = 0, prof_a1 = 0, prof_a2 = 0, prof_a3 = 0, prof_a4 = 0, prof_a5 = 0, prof_a6 = 0, prof_a7 = 0; \
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); prof_last = __builtin_amdgcn_readfirstlane((unsigned)t_); }
#define PHASE_MARK(i, reg) do { unsigned long long t_; asm volatile("v_cmp_eq_u32 vcc, %1, %1\n\ts_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(reg) : "memory", "vcc"); \
    const unsigned tl_ = __builtin_amdgcn_readfirstlane((unsigned)t_); prof_a##i = __builtin_amdgcn_readfirstlane(prof_a##i + (tl_ - prof_last)); prof_last = tl_; } while (0)
#define PHASE_DUMP do { if (threadIdx.x == 0 && blockIdx.x < 2048) { long long *o_ = g_phase_prof + blockIdx.x * 8; o_[0] = prof_a0; o_[1] = prof_a1; o_[2] = prof_a2; o_[3] = prof_a3; \
    o_[4] = prof_a4; o_[5] = prof_a5; o_[6] = prof_a6; o_[7] = prof_a7; } } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i, reg)
#define PHASE_DUMP
#endif
#include "mfma_bn.hpp"

namespace s5 {

// eight values in int32 registers -> byte planes (2 packed registers each), k order preserved
__device__ __forceinline__ void planes8_from_i32(const int32_t (&v)[8], v2i &hi, v2i &lo)
{
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned p01 = perm((unsigned)v[4 * j + 1], (unsigned)v[4 * j], 0x05010400u);
        const unsigned p23 = perm((unsigned)v[4 * j + 3], (unsigned)v[4 * j + 2], 0x05010400u);
        lo[j] = (int)(perm(p23, p01, 0x05040100u) ^ 0x80808080u);
        hi[j] = (int)perm(p23, p01, 0x07060302u);
    }
}

// Tiles of the per-layer kernels are 64 steps of ONE sequence inside a step range [t_lo, t_lo + t_len) (the
// whole sequence, or one chunk of the bproj | scan | cgate pipeline): tile -> (sequence, first step, valid steps)
struct StepRange {
    int32_t t_lo, t_len; // t_lo: a multiple of 64; t_len: any length >= 1 (a sequence's last 4-step block may be partial)
};
template <int FT = 64>
__device__ __forceinline__ void tile_of(int64_t tile, const StepRange &sr, int64_t &b, int &t, int &nvalid)
{
    const int tps = (sr.t_len + FT - 1) / FT;
    b = tile / tps;
    const int tt = (int)(tile - b * tps) * FT;
    t = sr.t_lo + tt;
    nvalid = sr.t_len - tt < FT ? sr.t_len - tt : FT;
}

// The same walk without a division per tile: a workgroup's tiles are gridDim.x apart, so (sequence, tile within it) advances
// by a fixed pair with one carry.  All wave-uniform scalar arithmetic (tile_of's 64-bit division by a run-time value is a
// software routine of some forty instructions, and the tile kernels called it two or three times per tile).
template <int FT = 64>
struct TileWalk {
    int tps, db, dti; // tiles per sequence; gridDim.x = db * tps + dti
    int b, ti;        // current tile: sequence b, tile ti of it
    __device__ __forceinline__ TileWalk(int64_t tile0, const StepRange &sr, unsigned step)
    {
        tps = (sr.t_len + FT - 1) / FT;
        db = (int)(step / (unsigned)tps); dti = (int)(step % (unsigned)tps);
        b = (int)((unsigned)tile0 / (unsigned)tps); ti = (int)((unsigned)tile0 % (unsigned)tps);
    }
    __device__ __forceinline__ void advance()
    {
        b += db; ti += dti;
        if (ti >= tps) { ti -= tps; ++b; }
    }
    __device__ __forceinline__ TileWalk next() const { TileWalk n = *this; n.advance(); return n; }
    __device__ __forceinline__ int t(const StepRange &sr) const { return sr.t_lo + ti * FT; }
    __device__ __forceinline__ int nvalid(const StepRange &sr) const { const int left = sr.t_len - ti * FT; return left < FT ? left : FT; }
};

// SM (stream mode) 0: int32 scan-native items; 1: int16 items (the host has checked that every value fits: Bu bits minus
// the shift to the state exponent <= 16; one item is then 8 bytes); 2: the pair kernel's K stream (scan_quad.hpp):
// K = (Bu << 16) + k in pair-native order, one 16-byte item per producer lane; 3: the LDS-fed pair kernel's int16 Bu
// stream (pair16-native), one 8-byte item per producer lane
// NC: 32-column tiles of [B_re | B_im] (= 2P / 32).  NC == NT: one tile per wave, both 32-frame halves of the tile's 64
// frames; NC == NT / 2 (a layer compacted to its live states, s5fxp_fast.hpp): wave w takes column tile w % NC and the ONE
// half w / NC -- the workgroup keeps its size, so phase A (the bulk of the kernel) is unchanged; NC == NT / 4 (a 128-state
// layer on 32 slots): the same, and the waves with w / NC >= 2 sit phase B out.
template <int KS, int NT, bool TRACE, int SM = 0, int NC = NT>
__global__ __launch_bounds__(64 * NT, 4) void k_bproj_p(BprojM2Args a, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(a.bn.dyn, g * go.ws); gshift(a.bn.xe.dyn, g * go.ws); gshift(a.x, g * go.ws); gshift(a.bq, g * go.ws); gshift(a.u, g * go.ws);
        gshift(a.ext, g * go.ws); gshift(a.status, g * go.status); gshift(a.status_exps, g * go.status);
    }
    static_assert(NC == NT || 2 * NC == NT || 4 * NC == NT, "column tiles per workgroup");
    constexpr int H = 32 * KS, FT = 64, KP = 32 * KS + 16, PC = 16 * NC, SUB0_STEP = NT / NC; // halves a wave takes: 2 / SUB0_STEP
    constexpr int VPF = H / 8;             // 16-byte vectors per frame
    constexpr int NTHR = 64 * NT;          // one wave per column tile: 256 threads at dim 0.5, 512 at dim 1.0
    constexpr int NV = FT * VPF / NTHR;    // vectors per thread and tile
    constexpr int NCT = 1;                 // column tiles per wave
    constexpr int PLANE = FT * KP;
    static_assert(FT * VPF % NTHR == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int32_t *tab = reinterpret_cast<int32_t *>(smem);             // 4*H BatchNorm operands
    int8_t *Xh = smem + 16 * H, *Xl = Xh + 2 * PLANE;             // [buf][frame][KP]
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int wct = wave % NC, wsub = wave / NC; // this wave's column tile and first 32-frame half
    const StepRange sr{a.t_lo, a.t_len};
    const int64_t tiles = (a.N / a.L) * ((sr.t_len + FT - 1) / FT);
    int64_t tile = blockIdx.x;

    v4i raw[NV];
    auto fetch = [&](const TileWalk<FT> &tw) {
        const int64_t b = tw.b;
        const int t = tw.t(sr), nv = tw.nvalid(sr);
        const char *xb = reinterpret_cast<const char *>(a.x + (b * a.L + t) * H); // wave-uniform; 32-bit byte offsets from here
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + NTHR * i;
            int f = v / VPF;
            f = f < nv ? f : nv - 1;
            raw[i] = *reinterpret_cast<const v4i *>(xb + 2u * (unsigned)(f * H + 8 * (v % VPF)));
        }
    };
    TileWalk<FT> walk(tile, sr, gridDim.x);
    if (tile < tiles) fetch(walk);
    // this wave's weight columns (B operand) and per-channel constants stay in registers
    v4i wreg[NCT][KS];
    int32_t csv[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int col = 32 * (wct + NC * c) + r;
        csv[c] = a.w.cs128[col];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wreg[c][ks] = *reinterpret_cast<const v4i *>(a.w.wt + (size_t)col * a.w.Kp + 32 * ks + 16 * h);
    }
#ifdef S5_BPROJ_CSR
    // EXPERIMENT build only (-DS5_BPROJ_CSR=<max nonzeros per column>; tools/variant.py): the
    // weight operand as compressed columns -- per output column the (k, w) pairs of its nonzeros, padded to the widest
    // column of the wave -- and the contraction on the VALU, one multiply-add per (frame, nonzero) and byte plane,
    // instead of the dense zero-filled MFMA operand.  Same accumulators, same epilogue, same results.
    constexpr int ELLW = S5_BPROJ_CSR;
    uint16_t *ell = reinterpret_cast<uint16_t *>(smem + 16 * H + 4 * PLANE); // [32 * NC columns][ELLW]: k | w << 8
    int *ellmax = reinterpret_cast<int *>(ell + 32 * NC * ELLW);            // [NC]: widest column of the tile
    if (threadIdx.x < NC) ellmax[threadIdx.x] = 0;
    __syncthreads();
    if (wsub == 0 && h == 0) {
        const int col = 32 * wct + r;
        int n = 0;
        for (int k = 0; k < H; ++k) {
            const int8_t wv = a.w.wt[(size_t)col * a.w.Kp + k];
            if (wv != 0 && n < ELLW) ell[col * ELLW + n++] = (uint16_t)(k | ((unsigned)(uint8_t)wv << 8));
        }
        for (int j = n; j < ELLW; ++j) ell[col * ELLW + j] = 0;
        atomicMax(ellmax + wct, n);
    }
#endif
    // The BatchNorm exponents of this layer: read from *dyn, or -- single-rank forwards -- derived here, by every
    // workgroup for itself, from the per-channel extremes the producer of the layer input left behind (the first tile's
    // loads are in flight meanwhile).  A single workgroup doing this at the tail of the producer kernel cost 4-7 us of
    // serialised round trips (atomics -> ticket -> loads -> arithmetic) on the critical path between two layers.
    const LayerDyn d = a.ext ? bn_finalize_mm_body(a.bn, a.ext, H, const_cast<LayerDyn *>(a.bn.dyn), a.status, a.status_exps, a.bn.xe.get(), a.ext_reps,
                                                   blockIdx.x == 0)
                             : *a.bn.dyn;
    const Bn16 bn = bn16_setup(a.bn, d, tab, H);
    __syncthreads();

    prologue_loads_done();
    for (int it = 0; tile < tiles; tile += gridDim.x, ++it, walk.advance()) {
        const int64_t b0 = walk.b;
        const int t0 = walk.t(sr), nvalid = walk.nvalid(sr);
        const int64_t n0 = b0 * a.L + t0;
        char *ub = reinterpret_cast<char *>(a.u + n0 * H); // wave-uniform base of this tile's rows of u
        int8_t *xh = Xh + (it & 1) * PLANE, *xl = Xl + (it & 1) * PLANE;
        // ---- phase A: BatchNorm chain, u, byte planes
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + NTHR * i, f = v / VPF, og = v % VPF;
            const int64_t n = n0 + f;
            int32_t xin[8], t[8], u[8];
            unpack8_i16(raw[i], xin);
            bn16_x4(bn, reinterpret_cast<const int32_t(&)[4]>(xin[0]), 8 * og, reinterpret_cast<int32_t(&)[4]>(t[0]),
                    reinterpret_cast<int32_t(&)[4]>(u[0]));
            bn16_x4(bn, reinterpret_cast<const int32_t(&)[4]>(xin[4]), 8 * og + 4, reinterpret_cast<int32_t(&)[4]>(t[4]),
                    reinterpret_cast<int32_t(&)[4]>(u[4]));
            if (f < nvalid && (TRACE || !a.no_u)) {
                const v2i p0 = pack4_i16(u[0], u[1], u[2], u[3]), p1 = pack4_i16(u[4], u[5], u[6], u[7]);
                *reinterpret_cast<v4i *>(ub + 2u * (unsigned)(f * H + 8 * og)) = v4i{p0[0], p0[1], p1[0], p1[1]};
                if (TRACE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (a.tr_pre_s5) a.tr_pre_s5[n * H + 8 * og + e] = t[e];
                        if (a.tr_u) a.tr_u[n * H + 8 * og + e] = u[e];
                    }
                }
            }
            v2i hi, lo;
            planes8_from_i32(u, hi, lo);
            *reinterpret_cast<v2i *>(xh + f * KP + 8 * og) = hi;
            *reinterpret_cast<v2i *>(xl + f * KP + 8 * og) = lo;
        }
        if (tile + gridDim.x < tiles) fetch(walk.next()); // in flight during phase B
        __syncthreads();
        // ---- phase B: this wave's column tile(s), both 32-frame halves
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = 32 * (wct + NC * c) + r;
            // SM >= 2: the weight columns are packed so that lanes r and r ^ 16 hold re and im of the SAME state
            // (pack_fast: bproj_pair); otherwise columns [0, P) are re, [P, 2P) im
            const int cc = SM >= 2 ? (r >> 4) : (col >= PC ? 1 : 0), p = SM >= 2 ? 16 * (wct + NC * c) + (r & 15) : col - cc * PC;
            const int rs = cc ? a.rs_im : a.rs_re, bits = cc ? a.bim_bits : a.bre_bits, sh = cc ? a.sh_im : a.sh_re;
            const int lsh = sh < 0 ? -sh : 0, rsh = sh > 0 ? sh : 0;
            SatB sbu; // this lane's clip bounds (re or im width), pinned in registers: not recomputed per element
            sbu.hi = (int32_t)((1u << (bits - 1)) - 1u); sbu.lo = ~sbu.hi;
            asm volatile("" : "+v"(sbu.lo), "+v"(sbu.hi));
#pragma unroll
            for (int sub0 = 0; sub0 < 2; sub0 += SUB0_STEP) {
                const int sub = sub0 + wsub;
                if (SUB0_STEP > 2 && sub >= 2) break; // a quarter of the column tiles: half of the waves have no unit in phase B
                const int8_t *rowh = xh + (32 * sub + r) * KP + 16 * h, *rowl = xl + (32 * sub + r) * KP + 16 * h;
                v16i acc;
#ifdef S5_BPROJ_CSR
                {
                    (void)rowh; (void)rowl;
                    int32_t ah[16], al[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) ah[i] = al[i] = 0;
                    const uint16_t *mine = ell + (32 * (wct + NC * c) + r) * ELLW;
                    const int nz = ellmax[wct + NC * c];
                    for (int j = 0; j < nz; ++j) {
                        const unsigned e = mine[j];
                        const int k = e & 0xff, wv = (int8_t)(e >> 8);
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int f = 32 * sub + 8 * (i >> 2) + 4 * h + (i & 3);
                            ah[i] += wv * xh[f * KP + k];
                            al[i] += wv * xl[f * KP + k];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = wadd(wadd(wshl(ah[i], 8), csv[c]), al[i]);
                }
#else
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i *>(rowh + 32 * ks), wreg[c][ks], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = wadd(wshl(acc[i], 8), csv[c]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i *>(rowl + 32 * ks), wreg[c][ks], acc, 0, 0, 0);
#endif
                // rows (frames) (i&3) + 8*(i>>2) + 4*h of this half: registers 4g..4g+3 are one 4-step block
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int o = 32 * sub + 8 * g + 4 * h;
                    // only the stores are guarded: with the arithmetic inside the branch the four groups of a lane run one
                    // after the other, each a short dependent chain, and three waves per SIMD do not hide that
                    const bool live = o < nvalid;
                    {
                        v4i q;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int32_t bu = sat(asr(acc[4 * g + e], rs), sbu);
                            q[e] = asr(wshl(bu, lsh), rsh);
                            if (TRACE && o + e < nvalid) {
                                if (!cc && a.tr_bu_re) a.tr_bu_re[(n0 + o + e) * PC + p] = bu;
                                if (cc && a.tr_bu_im) a.tr_bu_im[(n0 + o + e) * PC + p] = bu;
                            }
                        }
                        if (SM == 3) {
                            // the same item as SM == 2 (steps 0 and 2 from the other component's lane), as plain int16 Bu: the
                            // recurrence kernel's helper wave forms K.  h = 0 / 1 lanes hold the two blocks of a pair, so one
                            // wave store fills 512 contiguous bytes
                            const int32_t o0 = __builtin_amdgcn_ds_swizzle(q[0], 0x401f), o2 = __builtin_amdgcn_ds_swizzle(q[2], 0x401f);
                            const v2i item = pack4_i16(o0, o2, q[1], q[3]);
                            // pair16_half(b0, (t0 + o) >> 2, p, cc): the tile's first block of state group 0 is the uniform base
                            if (live && (a.live_slots <= 0 || p < a.live_slots)) {
                                char *qb = reinterpret_cast<char *>(reinterpret_cast<int16_t *>(a.bq) + pair16_half(b0, t0 >> 2, 0, 0, a.TB, PC));
                                const unsigned qo = 2u * (unsigned)((((((p >> 5) * (a.TB >> 1) + (o >> 3)) << 6) + 2 * (p & 31) + cc) * 8) + 4 * ((o >> 2) & 1));
                                *reinterpret_cast<v2i *>(qb + qo) = item;
                            }
                        } else if (SM == 2) {
                            // K = (Bu << 16) + k.  Pair-native items: lane A = [Kim0 Kim2 Kre1 Kre3], lane B = [Kre0 Kre2
                            // Kim1 Kim3]: steps 0 and 2 come from the OTHER component's lane (r ^ 16, ds_swizzle), steps 1
                            // and 3 are this lane's own; the re lane writes item A, the im lane item B -- one 16-byte store
                            // each, 512 contiguous bytes per half wave
                            const int32_t kc = cc ? 0 : a.k_re;
                            const int32_t k0 = wadd(wshl(q[0], 16), kc), k2 = wadd(wshl(q[2], 16), kc);
                            v4i item;
                            item[0] = __builtin_amdgcn_ds_swizzle(k0, 0x401f);
                            item[1] = __builtin_amdgcn_ds_swizzle(k2, 0x401f);
                            item[2] = wadd(wshl(q[1], 16), kc);
                            item[3] = wadd(wshl(q[3], 16), kc);
                            if (live) *reinterpret_cast<v4i *>(a.bq + pair_word(b0, (t0 + o) >> 2, p, a.TB, PC) + 4 * cc) = item;
                        } else if (SM == 1) {
                            if (live && (a.live_slots <= 0 || p < a.live_slots))
                                *reinterpret_cast<v2i *>(reinterpret_cast<int16_t *>(a.bq) + native_word(b0, t0 + o, p, cc, a.TB, PC)) =
                                    pack4_i16(q[0], q[1], q[2], q[3]);
                        } else if (live)
                            *reinterpret_cast<v4i *>(a.bq + native_word(b0, t0 + o, p, cc, a.TB, PC)) = q;
                    }
                }
            }
        }
    }
}

// uniform operands of a change_cfg (fxp_prims.hpp chcfg) applied to many elements; `on` false = identity
using v2u16 = __attribute__((ext_vector_type(2))) unsigned short;

struct CfgOp {
    int l, r, b;
    SatB sb; // the clip bounds in VGPRs (fxp_prims.hpp sat_bounds): make_cfg is called once per kernel
    __device__ __forceinline__ int32_t operator()(int32_t d) const { return sat(asr(wshl(d, l), r), sb); }
};
__device__ __forceinline__ CfgOp make_cfg(bool on, int bits, int e, int bits2, int e2)
{
    CfgOp c;
    c.l = on && e2 > e ? e2 - e : 0;
    c.r = on && e > e2 ? e - e2 : 0;
    const int b1 = on && e2 != e ? bits : 32, b2 = on && bits > bits2 ? bits2 : 32;
    c.b = b1 < b2 ? b1 : b2;
    c.sb = sat_bounds(c.b);
    return c;
}

// two-plane MFMA, A operand (weights, rows = channels) in registers, B operand (byte planes) from LDS:
// lane = frame, registers = channels (i&3) + 8*(i>>2) + 4*(lane>>5) of the 32-channel tile
template <int KSTEPS>
__device__ __forceinline__ void mfma_planes(v16i &acc, const v4i (&w)[KSTEPS], const int8_t *rowh, const int8_t *rowl,
                                            const int32_t *cs)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks], *reinterpret_cast<const v4i *>(rowh + 32 * ks), acc, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v4i c = *reinterpret_cast<const v4i *>(cs + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * g + e] = wadd(wshl(acc[4 * g + e], 8), c[e]);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks], *reinterpret_cast<const v4i *>(rowl + 32 * ks), acc, 0, 0, 0);
}

// the same for NPL byte planes [plane][frame][k] (plane NPL-1 = signed top byte), Horner from the top: the constant
// 128*sum(w) enters at every shift, so after NPL-1 shifts it has the weight 2^(8(NPL-2)) + ... + 2^8 + 1 that the
// +128 offsets of the lower planes need
#ifdef S5_GATE_CHECK
// EXPERIMENT builds only (tools/variant.py): the test an activation-gating kernel makes before each
// MFMA of the gate kernel's two projections -- is this wave's operand fragment (32 frames x 32 k, every byte plane) all zero?
//   -DS5_GATE_CHECK=1 (gate_count): counts them; the MFMA still runs.  [0]/[1]: fragments / all-zero fragments of the C
//     projection's operand (the states after the complex ReLU, fxpmodel.py:740-742), [2]/[3]: the same for out2's operand (the
//     SSM output after the ReLU, :1125).  Not for timing: every wave hits the same two counters.
//   -DS5_GATE_CHECK=2 (gate_skip): branches around the MFMAs of an all-zero fragment, no counters: what a gating kernel
//     costs.  (Skipping drops the fragment's share of the lower planes' +128 correction: exact only when nothing is skipped
//     or the skipped k carry zero weights, as the padding slots of a compacted layer do.)
__device__ unsigned long long g_gate_frag[4];
template <int NPL>
__device__ __forceinline__ bool gate_check(const int8_t *frag0, int plane_stride, int slot)
{
    unsigned nz = 0;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        const v4i b = *reinterpret_cast<const v4i *>(frag0 + pl * plane_stride);
        const unsigned pat = pl == NPL - 1 ? 0u : 0x80808080u; // the lower planes hold (byte ^ 0x80)
        nz |= ((unsigned)b[0] ^ pat) | ((unsigned)b[1] ^ pat) | ((unsigned)b[2] ^ pat) | ((unsigned)b[3] ^ pat);
    }
    const bool any = __any(nz != 0);
#if S5_GATE_CHECK == 1
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_gate_frag[slot], 1ull);
        if (!any) atomicAdd(&g_gate_frag[slot + 1], 1ull);
    }
    return true;
#else
    return any;
#endif
}
#endif
template <int KSTEPS, int NPL>
__device__ __forceinline__ void mfma_nplanes(v16i &acc, const v4i (&w)[KSTEPS], const int8_t *row0, int plane_stride,
                                             const int32_t *cs)
{
#ifdef S5_GATE_CHECK
    bool live[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) live[ks] = gate_check<NPL>(row0 + 32 * ks, plane_stride, 0);
#endif
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl) {
        if (pl != NPL - 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i c = *reinterpret_cast<const v4i *>(cs + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * g + e] = wadd(wshl(acc[4 * g + e], 8), c[e]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#ifdef S5_GATE_CHECK
            if (!live[ks]) continue;
#endif
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[ks], *reinterpret_cast<const v4i *>(row0 + pl * plane_stride + 32 * ks), acc, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Encoder, phase-split: x int32 (N,K) -> relu(dense) int16 (N,H).  fxpmodel.py:331-366, 1263-1266.
// Six waves, 64-frame tiles.  Phase A: a wave reads whole rows (64 lanes x 4 consecutive k, 1 KB contiguous)
// plus the K-256 tail, converts, and writes byte planes [frame][304]; phase B: wave (half, column tile).
// ext != nullptr: the per-channel extremes of the output (layer 0's BatchNorm operand, mfma_bn.hpp) are gathered
// on the way -- a lane keeps (max, 65535 - min) of its 16 channels as packed u16 pairs (the output is >= 0 after
// the ReLU); the atomics are spread over ext_reps replicas (mfma_bn.hpp EXT_REPS).
// LDS: [cs128 Np][bias_eff Np][X hi][X lo][ext hi H][ext lo H]
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(384, 3) void k_enc_p(EncArgs a, float *ext, int ext_reps, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(a.x, g * go.x); gshift(a.y, g * go.ws); gshift(a.status, g * go.status); gshift(ext, g * go.ws);
    }
    constexpr int KS = 9, FT = 64, KP = 32 * KS + 16, NW = 6, H = 32 * NT;
    constexpr int NU = 2 * NT / NW, SUBSTEP = NW / NT;
    constexpr int RPW = (FT + NW - 1) / NW; // rows per wave
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int32_t *cs = reinterpret_cast<int32_t *>(smem), *be = cs + H;
    int8_t *Xh = reinterpret_cast<int8_t *>(be + H), *Xl = Xh + FT * KP;
    uint32_t *ehi = reinterpret_cast<uint32_t *>(Xl + FT * KP), *elo = ehi + H;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int ct = wave % NT, sub0 = wave / NT, ch0 = 32 * ct + 4 * h;
    const int64_t tiles = (a.N + FT - 1) / FT;
    const int K = a.K, rem = K - 256;
    uint32_t pk[16]; // low half: max, high half: 65535 - min
#pragma unroll
    for (int i = 0; i < 16; ++i) pk[i] = 0;
    v4i wreg[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        wreg[ks] = *reinterpret_cast<const v4i *>(a.w.wt + (size_t)(32 * ct + r) * a.w.Kp + 32 * ks + 16 * h);
    for (int i = threadIdx.x; i < H; i += 384) {
        cs[i] = a.w.cs128[i];
        be[i] = a.bias_eff[i];
        ehi[i] = 0;
        elo[i] = 0;
    }
    const CfgOp cv = make_cfg(a.conv != 0, a.xb, a.xe, a.inp_bits, a.inp_exp);
    const SatB so = sat_bounds(a.out_bits);
    // rows wave, wave+6, ... of the tile.  Two workgroups of six waves per CU are three waves per SIMD whatever the kernel
    // does, so it may hold 168 registers: all of a tile's rows are requested a tile ahead (dim 0.5; the two-unit phase B
    // of dim 1.0 has no room for that: there the first RA rows are prefetched and the rest requested at the top of phase A).
    // The prefetch is issued behind the compiler's back (scan_quad.hpp vm_wait): its own wait at the first use -- a tile
    // later, behind phase B's stores -- would be vmcnt(0), every tile opening with a wait for the previous tile's stores.
    constexpr int RA = NT <= 3 ? RPW : 3, RB = RPW - RA;
    v4i rawa[RA], rawb[RB > 0 ? RB : 1];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto row_base = [&](int64_t tl, int i) { // wave-uniform
        int64_t n = tl * FT + wave_u + NW * i;
        n = n < a.N ? n : a.N - 1;
        return reinterpret_cast<const char *>(a.x + n * K);
    };
    auto row_ptr = [&](int64_t tl, int i) {
        return reinterpret_cast<const v4i *>(row_base(tl, i) + 16 * l); // 4-byte aligned 16-byte load
    };
    auto convert_row = [&](const v4i &q, int f, bool &wide) {
        int32_t v[4] = {q[0], q[1], q[2], q[3]};
        if (a.conv) { // uniform: usually the input already has the encoder's configuration
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = cv(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) wide |= (v[e] != (int32_t)(int16_t)v[e]);
        const unsigned p01 = perm((unsigned)v[1], (unsigned)v[0], 0x05010400u), p23 = perm((unsigned)v[3], (unsigned)v[2], 0x05010400u);
        *reinterpret_cast<int32_t *>(Xl + f * KP + 4 * l) = (int32_t)(perm(p23, p01, 0x05040100u) ^ 0x80808080u);
        *reinterpret_cast<int32_t *>(Xh + f * KP + 4 * l) = (int32_t)perm(p23, p01, 0x07060302u);
    };
    int64_t tile = blockIdx.x;
    if (tile < tiles) {
#pragma unroll
        for (int i = 0; i < RA; ++i) rawa[i] = gload16_hidden(row_base(tile, i), 16u * (unsigned)l);
    }
    bool wide = false;
    __syncthreads();
    PHASE_DECL
    prologue_loads_done();
    const bool even = a.M == H; // no ragged column tile: full tiles store unconditionally
    for (; tile < tiles; tile += gridDim.x) {
        const int64_t n0 = tile * FT;
        PHASE_MARK(0, l); // loop top (includes the previous tile's closing barrier)
        // ---- phase A
        // the prefetched rows are older than the previous tile's stores: NU x 4 per wave on the unconditional path (the
        // conditional one ends with a full wait)
        vm_wait<4 * NU>(rawa);
        if constexpr (RB > 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i) rawb[i] = *row_ptr(tile, RA + i);
        }
#pragma unroll
        for (int i = 0; i < RA; ++i)
            if (wave_u + NW * i < FT) convert_row(rawa[i], wave_u + NW * i, wide);
        PHASE_MARK(1, rawa[RA - 1][0]); // prefetched rows converted
        if constexpr (RB > 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i)
                if (wave_u + NW * (RA + i) < FT) convert_row(rawb[i], wave_u + NW * (RA + i), wide);
        }
        PHASE_MARK(2, rawb[0][0]); // rows requested at the top (their HBM latency included)
        for (int e = threadIdx.x; e < FT * rem; e += 384) { // the K-256 tail of every row
            const int f = e / rem, k = 256 + e % rem;
            int64_t n = n0 + f;
            n = n < a.N ? n : a.N - 1;
            const int32_t v = cv(a.x[n * K + k]);
            wide |= (v != (int32_t)(int16_t)v);
            Xl[f * KP + k] = (int8_t)((v & 0xff) ^ 0x80);
            Xh[f * KP + k] = (int8_t)(v >> 8);
        }
        if (tile + gridDim.x < tiles) {
#pragma unroll
            for (int i = 0; i < RA; ++i) rawa[i] = gload16_hidden(row_base(tile + gridDim.x, i), 16u * (unsigned)l); // in flight during phase B
        }
        PHASE_MARK(3, l); // tail column + prefetch issue
        __syncthreads();
        PHASE_MARK(4, l); // mid barrier
        // ---- phase B
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int sub = sub0 + u * SUBSTEP;
            const int64_t n = n0 + 32 * sub + r;
            v16i acc;
            mfma_planes<KS>(acc, wreg, Xh + (32 * sub + r) * KP + 16 * h, Xl + (32 * sub + r) * KP + 16 * h, cs + ch0);
            PHASE_MARK(5, acc[15]); // operand reads + MFMA chain, complete
            auto group = [&](int g) {
                const int ch = ch0 + 8 * g;
                const v4i bv = *reinterpret_cast<const v4i *>(be + ch);
                int32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int32_t v = sat(asr(acc[4 * g + e], a.rs), so);
                    v = sat(wadd(v, bv[e]), so);
                    o[e] = v < 0 ? 0 : v;
                    // (v, 65535 - v) as a u16 pair; one packed max keeps both running extremes
                    const uint32_t t = (uint32_t)__umul24((unsigned)o[e], 0x10001u) ^ 0xffff0000u;
                    pk[4 * g + e] = __builtin_bit_cast(
                        uint32_t, __builtin_elementwise_max(__builtin_bit_cast(v2u16, pk[4 * g + e]), __builtin_bit_cast(v2u16, t)));
                }
                *reinterpret_cast<v2i *>(a.y + n * a.M + ch) = pack4_i16(o[0], o[1], o[2], o[3]);
            };
            if (even && n0 + FT <= a.N) { // no control flow around the stores (see vm_wait above)
#pragma unroll
                for (int g = 0; g < 4; ++g) group(g);
            } else {
                if (n < a.N) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (ch0 + 8 * g < a.M) group(g);
                }
                prologue_loads_done(); // nothing in flight behind a conditional store
            }
        }
        PHASE_MARK(6, pk[15]); // epilogue arithmetic done, stores issued
        __syncthreads(); // planes are single-buffered
    }
    PHASE_DUMP;
    if (__any(wide) && l == 0) atomicOr(a.status, ST_WIDE_INPUT);
    if (!ext) return;
    // ---- extremes: fold the 32 frame lanes of each half wave, then the waves of the workgroup (LDS), then one
    // atomic per channel and bound
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        uint32_t v = pk[i];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
            const uint32_t lo16 = (v & 0xffffu) > (w & 0xffffu) ? (v & 0xffffu) : (w & 0xffffu);
            const uint32_t hi16 = (v >> 16) > (w >> 16) ? (v >> 16) : (w >> 16);
            v = lo16 | (hi16 << 16);
        }
        const int ch = ch0 + 8 * (i >> 2) + (i & 3);
        if (r == 0 && ch < a.M) {
            atomicMax(&ehi[ch], v & 0xffffu);
            atomicMax(&elo[ch], v >> 16);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.M) { // max = ehi, min = 65535 - elo; as the positive floats of mfma_bn.hpp
        const int c = threadIdx.x;
        uint32_t *dst = reinterpret_cast<uint32_t *>(ext) + (ext_reps > 1 ? (int)(blockIdx.x % ext_reps) : 0) * 2 * a.M;
        atomicMax(dst + c, __float_as_uint(EXT_BIAS - (float)(65535 - (int)elo[c])));
        atomicMax(dst + a.M + c, __float_as_uint(EXT_BIAS + (float)ehi[c]));
    }
}

// ---------------------------------------------------------------------------------------------
// where masked-off lanes of a store send their value instead (never read)
__device__ int32_t g_store_sink[64];
// Decoder, phase-split: h int16 (N,H) -> dense int32 (N,M), M <= 288.  fxpmodel.py:331-366, 1272-1274.
// Six waves, 64-frame tiles.  Phase B runs as D = X * W (lane = output column, registers = frames): every
// store instruction writes 128 contiguous bytes of one output row per half wave.
// LDS: [X hi][X lo]
// ---------------------------------------------------------------------------------------------
// RESID: the last layer's residual pass (mfma_bn.hpp k_resid_minmax16 without its extremes, which nobody needs after the
// last layer) happens here, on the way in: a.x is the layer's INPUT (skip), rz.z the gate kernel's output, and
// h = relu(z + skip) (fxpmodel.py:1147-1159) is formed in registers -- the h plane is neither written nor read back
// (-75 + 25 MB per batch at configs[1]) and the forward is one launch shorter.  The residual exponents come from the
// maxima the gate kernel left, derived by every workgroup for itself as in the B projection.
struct DecResid {
    const int16_t *z;
    ResidHead hd;
    int32_t res_bits, skip_bits;
};
template <int KS, bool RESID = false>
__global__ __launch_bounds__(384, KS == 6 && RESID ? 2 : 3) void k_dec_p(DecArgs a, DecResid rz, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(a.x, g * go.ws); gshift(a.y, g * go.y); gshift(a.xe.dyn, g * go.ws); gshift(a.status, g * go.status);
        if constexpr (RESID) {
            gshift(rz.z, g * go.ws); gshift_nn(rz.hd.d, g * go.ws); gshift(rz.hd.skip_e.dyn, g * go.ws);
            gshift_nn(rz.hd.status_exps, g * go.status);
        }
    }
    constexpr int H = 32 * KS, FT = 64, KP = H + 16, NW = 6, CT = 9, CPW = 3;
    constexpr int VPF = H / 8, NV = FT * VPF / 384;
    static_assert(FT * VPF % 384 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    int8_t *Xh = smem, *Xl = Xh + FT * KP;
    const int l = threadIdx.x & 63, r = l & 31, h = l >> 5, wave = threadIdx.x >> 6;
    const int sub = wave / 3, c0 = wave % 3;
    const int64_t tiles = (a.N + FT - 1) / FT;
    AddCb rp{};
    if constexpr (RESID) {
        __shared__ AddCb sp;
        if (rz.hd.enable) {
            if (threadIdx.x == 0) {
                sp = finalize_add_cb(rz.hd.d->mx + (rz.hd.d->redo ? rz.hd.redo_slot : 8), rz.hd.res_exp, rz.hd.skip_e.get(), rz.res_bits, a.status);
                if (blockIdx.x == 0) {
                    rz.hd.d->res = sp;
                    rz.hd.status_exps[4] = sp.eo;
                }
            }
            __syncthreads();
            rp = sp;
        } else {
            rp = rz.hd.d->res;
        }
    }
    const int xe0 = RESID ? rp.eo : a.xe.get();
    const bool conv = a.xb > a.inp_bits || xe0 > a.inp_exp;
    int rs = (conv ? a.inp_exp : xe0) + a.w_exp - a.out_exp;
    if (rs < 0 || rs > 31) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(a.status, ST_NEGSHIFT);
        rs = rs < 0 ? 0 : 31;
    }
    const CfgOp cv = make_cfg(conv, a.xb, xe0, a.inp_bits, a.inp_exp);
    const SatB so = sat_bounds(a.out_bits);
    AddCbV rpv{};
    if constexpr (RESID) rpv = make_add_cb_v(rp, rz.res_bits, rz.skip_bits, rz.res_bits);
    v4i wreg[CPW][KS];
    int32_t csv[CPW], bev[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int col = 32 * (c0 + 3 * c) + r;
        csv[c] = a.w.cs128[col];
        bev[c] = a.bias_eff[col];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            wreg[c][ks] = *reinterpret_cast<const v4i *>(a.w.wt + (size_t)col * a.w.Kp + 32 * ks + 16 * h);
    }
    v4i raw[NV], rawz[RESID ? NV : 1];
    auto fetch = [&](int64_t tl) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + 384 * i;
            const int64_t left = a.N - tl * FT; // frames from the tile's first to the end of the tensor (wave-uniform)
            int f = v / VPF;
            f = f < left ? f : (int)left - 1;
            if constexpr (RESID)
                rawz[i] = gload16_hidden(reinterpret_cast<const char *>(rz.z + tl * FT * H), 2u * (unsigned)(f * H + 8 * (v % VPF)));
            // issued behind the compiler's back (see vm_wait): its wait-count pass would otherwise guard the first use of
            // these registers, a tile later, with vmcnt(0) -- behind the 48 stores of this tile's phase B
            raw[i] = gload16_hidden(reinterpret_cast<const char *>(a.x + tl * FT * H), 2u * (unsigned)(f * H + 8 * (v % VPF)));
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < tiles) fetch(tile);
    prologue_loads_done();
    for (; tile < tiles; tile += gridDim.x) {
        const int64_t n0 = tile * FT;
        // the prefetched rows are older than the previous tile's 3 x 16 stores per wave (every tile but the tensor's last is
        // full and stores unconditionally; that last one has no successor)
        vm_wait<3 * 16>(raw);
        if constexpr (RESID) vm_wait<3 * 16>(rawz);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + 384 * i, f = v / VPF, og = v % VPF;
            int32_t x[8];
            unpack8_i16(raw[i], x);
            if constexpr (RESID) {
                int32_t z[8];
                unpack8_i16(rawz[i], z);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int32_t rr = add_cb_apply(z[e], x[e], rpv);
                    x[e] = rr < 0 ? 0 : rr;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = cv(x[e]);
            v2i hi, lo;
            planes8_from_i32(x, hi, lo);
            *reinterpret_cast<v2i *>(Xh + f * KP + 8 * og) = hi;
            *reinterpret_cast<v2i *>(Xl + f * KP + 8 * og) = lo;
        }
        if (tile + gridDim.x < tiles) fetch(tile + gridDim.x);
        __syncthreads();
        const int8_t *rowh = Xh + (32 * sub + r) * KP + 16 * h, *rowl = Xl + (32 * sub + r) * KP + 16 * h;
        const int64_t nb = n0 + 32 * sub + 4 * h; // frame of accumulator register 0
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int col = 32 * (c0 + 3 * c) + r;
            v16i acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i *>(rowh + 32 * ks), wreg[c][ks], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = wadd(wshl(acc[i], 8), csv[c]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i *>(rowl + 32 * ks), wreg[c][ks], acc, 0, 0, 0);
            // Stores without control flow around them on full tiles (all but the tensor's last): lanes of the ragged last
            // column tile (col >= M) write to a sink word instead of being masked off.  Every conditional store would make
            // the number of memory operations in flight unknowable to the compiler's wait-count pass, and the wait it
            // then puts at the top of the next tile -- for the rows prefetched BEFORE these stores -- degenerates to
            // vmcnt(0): every tile would begin by waiting for the previous tile's stores to be acknowledged.
            const bool okc = col < a.M;
            char *yl = okc ? reinterpret_cast<char *>(a.y + n0 * a.M) + 4u * (unsigned)((32 * sub + 4 * h) * a.M + col)
                           : reinterpret_cast<char *>(as_global(&g_store_sink[l]));
            const unsigned ystep = okc ? 4u * (unsigned)a.M : 0u;
            if (n0 + FT <= a.N) {
                char *yp = yl; // a running pointer: sixteen hoisted offsets per column tile would cost the kernel its occupancy
#pragma unroll
                for (int i = 0; i < 16; ++i) { // frames (i & 3) + 8 * (i >> 2)
                    const int32_t v = sat(asr(acc[i], rs), so);
                    *reinterpret_cast<int32_t *>(yp) = sat(wadd(v, bev[c]), so);
                    yp += (i & 3) == 3 ? 5 * ystep : ystep;
                }
            } else {
                char *yp = yl;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int fo = (i & 3) + 8 * (i >> 2);
                    if (nb + fo < a.N) {
                        const int32_t v = sat(asr(acc[i], rs), so);
                        *reinterpret_cast<int32_t *>(yp) = sat(wadd(v, bev[c]), so);
                    }
                    yp += (i & 3) == 3 ? 5 * ystep : ystep;
                }
                prologue_loads_done(); // the tensor's last tile: nothing is left in flight on this path
            }
            __builtin_amdgcn_sched_barrier(0); // one column tile at a time: interleaved, the three of a wave do not fit its registers
        }
        __syncthreads(); // planes are single-buffered
    }
}

} // namespace s5
