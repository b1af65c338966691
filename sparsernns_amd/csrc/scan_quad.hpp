// scan_quad.hpp -- the exact sequential diagonal-SSM recurrence at 3 dependent VALU ops per step.
//
// Reference step (sparseRNNs/fxpmodel.py:155-169), per state, int32 with wrap, no clip:
//     re' = asr(Ar*xr, e_re) - asr(Ai*xi, e_re) + bre          im' = asr(Ar*xi, e_im) + asr(Ai*xr, e_im) + bim
// The recurrence is latency bound (L sequential steps, only B*P independent chains), so the kernel
// minimises the dependent chain, not the instruction count.  Four lanes (a quad) carry one state:
//
//   z  = c * x + k          v_mad_i32_i24   c = +-A * 2^(16-e)  (so the floor shift becomes ">> 16"),
//                                           k = 2^16 - 2^(16-e) on the lane that must deliver -floor(.)
//   q  = (z >> 16) + b      v_add_u32 SDWA  src0 = sign-extended high half of z; b = Bu on one lane of
//                                           each partner pair, 0 on the other
//   x' = q + q[partner]     v_add_u32 DPP   quad_perm; afterwards BOTH partners hold the new component
//
// Which lanes are partners alternates every step (phase A / phase B) so that every lane already holds
// the component its next multiply needs -- no cross-lane move in front of the multiply:
//   phase A: lanes {0,3} hold re, {1,2} hold im;  pairs (0,2)->re', (1,3)->im';  quad_perm [2,3,0,1]
//   phase B: lanes {0,2} hold re, {1,3} hold im;  pairs (0,3)->re', (1,2)->im';  quad_perm [3,2,1,0]
// Lane 0 always ends a step holding re', lane 1 im'.
//
// Exactness: with |c*x| + k < 2^31 the 24-bit multiply-add is the true integer, its ">>16" equals
// asr(A*x, e) (or its exact negation), and all additions wrap like int32.  The bound is
// |x| <= xmax = (2^31 - 1 - 2^16) / max|c|  (32767 for 16-bit Lambda at exponent 15, i.e. the state's
// nominal width).  The consumer of the states checks |x| <= xmax on every stored state; by
// induction over t that proves every product was exact.  If the check fails the forward re-runs the
// layer with the 32-bit one-lane-per-state kernel (k_scan_lane), so results are exact either way.
//
// Streams use the "scan-native" layout, written by the B projection and read by the C projection:
//     word(b, tb, p, c, j) = ((((b*TB + tb)*P + p)*2 + c)*4 + j),   t = 4*tb + j,  c: 0 = re, 1 = im
// so lanes 0/1 of a quad move 16-byte vectors of four consecutive steps and a wave (16 states) reads
// and writes 512 contiguous bytes per 4 steps.  Lanes 2/3 use an out-of-range buffer offset: their
// loads return 0 and their stores are dropped by the buffer range check.
#pragma once
#include "fxp_prims.hpp"

namespace s5 {

using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;
using i32x4 = __attribute__((ext_vector_type(4))) int;

// Grouped launches: one launch of a fused-path kernel may cover several independent reference batches ("groups": every
// group is its own compute_best batch with its own exponents, status words and streaming carry).  gridDim.y = number of
// groups, blockIdx.y = this workgroup's group; group g's tensors lie g * stride bytes behind group 0's -- the model input
// and output, the per-forward workspace (every activation, stream, LayerDyn and extremes buffer lives in it), the
// status words and the two carry arrays.  Kernels shift their pointers once, at the top.
struct GroupOff {
    int64_t x, y, ws, status, state_in, state_out; // byte strides
};
// Every pointer a kernel receives is device memory.  The shift goes through an explicit GLOBAL (address space 1) pointer:
// with integer arithmetic on the pointer value instead (ptrtoint / inttoptr), or with pointers that arrive inside a large
// by-reference argument struct, the compiler cannot prove the address space and emits FLAT loads and stores -- which count
// on lgkmcnt as well as vmcnt, so that every wait for LDS (each barrier of the tile kernels) also waits for the global
// prefetches in flight.
using gchar = __attribute__((address_space(1))) char;
template <class P> // P: any (possibly restrict-qualified) pointer type
__device__ __forceinline__ P as_global(P p)
{
    return (P)(char *)(gchar *)(char *)p;
}
template <class P> // a null pointer stays null
__device__ __forceinline__ void gshift(P &p, int64_t bytes)
{
    p = p ? (P)(char *)((gchar *)(char *)p + bytes) : (P) nullptr;
}
template <class P> // for pointers that are never null
__device__ __forceinline__ void gshift_nn(P &p, int64_t bytes)
{
    p = (P)(char *)((gchar *)(char *)p + bytes);
}

// Placed in front of a tile loop: everything the prologue requested (the weights that stay in registers, the first tile)
// has arrived.  Without it the compiler's wait-count pass has to assume, in EVERY iteration, that the prologue's loads
// may still be pending at the first use of a weight register, and the wait it inserts there (vmcnt counts in order) also
// waits for the newest loads in flight -- the next tile's prefetch, issued a few instructions earlier.
__device__ __forceinline__ void prologue_loads_done() { __builtin_amdgcn_s_waitcnt(0x0F70); } // vmcnt(0) only

// A 16-byte global load the compiler does not know to be a load (wave-uniform base + 32-bit byte offset), and the wait that
// makes its result usable: at most NEWER memory operations of this wave issued after it may still be in flight (vmcnt
// retires in order).  The registers pass through the wait statement, so nothing can be scheduled to read them above it.
// For prefetches whose first use lies behind conditional or numerous stores: there the wait the compiler would insert by
// itself is vmcnt(0).
using v4i_ = __attribute__((ext_vector_type(4))) int;
__device__ __forceinline__ v4i_ gload16_hidden(const char *base, unsigned off)
{
    v4i_ r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
__device__ __forceinline__ __attribute__((ext_vector_type(2))) int gload8_hidden(const char *base, unsigned off)
{
    __attribute__((ext_vector_type(2))) int r;
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(r) : "v"(off), "s"(base) : "memory");
    return r;
}
template <int NEWER, class T, int N>
__device__ __forceinline__ void vm_wait(T (&regs)[N])
{
    static_assert(NEWER >= 0 && NEWER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
    // volatile statements keep their order: every later reader of a register depends on its pass through here
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(regs[i]));
}
template <int NEWER, class T, int N, int M>
__device__ __forceinline__ void vm_wait(T (&regs)[N][M])
{
    static_assert(NEWER >= 0 && NEWER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NEWER) : "memory");
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) asm volatile("" : "+v"(regs[i][j]));
}

// word index of (sequence b, step t, state p, component c) in a scan-native stream with TB blocks/sequence
__device__ __forceinline__ int64_t native_word(int64_t b, int t, int p, int c, int TB, int P)
{
    return ((((b * TB + (t >> 2)) * P + p) * 2 + c) << 2) + (t & 3);
}

struct ScanQuadArgs {
    const int32_t *bq;          // native stream: Bu already shifted to the state exponent
    int32_t *xs;                // native stream: raw states
    const int32_t *a_re, *a_im; // (P)
    int32_t B, TB, P;           // TB = number of 4-step time blocks per sequence (stream extent)
    int32_t ea_re, ea_im;
    int32_t tb0, ntb;           // k_scan_quad_asm: first time block and block count of this launch (0, 0 = all)
    const int32_t *run_if;      // k_scan_quad32_asm: do the work only when *run_if != 0 (nullptr: always)
    const int32_t *x0_re, *x0_im; // (B,P) state before the first step (streaming carry), nullptr = zeros
    int32_t live_slots;         // k_scan_quad_asm16, > 0: state slots at or above it are not in the streams (ScanPairLArgs)
};

// the state a quad lane holds before an even step: lanes 0,3 the real part, lanes 1,2 the imaginary part
__device__ __forceinline__ int32_t quad_x0(const ScanQuadArgs &a, int b, int p, int r)
{
    if (!a.x0_re) return 0;
    return (r == 0 || r == 3) ? a.x0_re[(size_t)b * a.P + p] : a.x0_im[(size_t)b * a.P + p];
}

// PRE is "s_nop 1\n\t" for the first step after a 16-byte buffer_store: gfx940+ needs 2 wait states
// between such a store and a VALU write to one of its data registers, and hipcc pads nothing
// around inline asm.
#define S5_SCAN_STEP(PRE, PERM, XIN, XOUT, C, K, BQ)                                                           \
    asm volatile(PRE "v_mad_i32_i24 %0, %2, %3, %4\n\t"                                                          \
                 "v_add_u32_sdwa %0, sext(%0), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 "        \
                 "src1_sel:DWORD\n\t"                                                                          \
                 "s_nop 1\n\t"                                                                                 \
                 "v_add_u32_dpp %1, %0, %0 quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"                  \
                 : "=&v"(tmp), "=v"(XOUT)                                                                      \
                 : "v"(C), "v"(XIN), "v"(K), "v"(BQ))

// One wave = 16 states of one sequence.  DEPTH = time blocks (of 4 steps) kept in flight.
template <int DEPTH>
__global__ __launch_bounds__(256) void k_scan_quad(ScanQuadArgs a)
{
    const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
    // wave-uniform by construction; readfirstlane lets the compiler keep the descriptors in SGPRs
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave_in_block);
    const int groups = a.P >> 4; // waves per sequence
    const int b = wave / groups, p0 = (wave % groups) << 4;
    if (b >= a.B) return;
    const int s = lane >> 2, r = lane & 3;
    const int p = p0 + s;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int sre = 16 - a.ea_re, sim = 16 - a.ea_im;
    const int32_t kre = (1 << 16) - (1 << sre);
    // phase A: l0 (Ar->re) l1 (Ar->im) l2 (-Ai->re, k) l3 (Ai->im);  phase B: l2 (Ai->im), l3 (-Ai->re, k)
    int32_t cA, cB, kA = 0, kB = 0;
    if (r == 0) { cA = cB = Ar << sre; }
    else if (r == 1) { cA = cB = Ar << sim; }
    else if (r == 2) { cA = -(Ai << sre); kA = kre; cB = Ai << sim; }
    else { cA = Ai << sim; cB = -(Ai << sre); kB = kre; }

    // per-wave buffer descriptors: base = first word of (b, tb=0, p0); lanes 2,3 are out of range
    const size_t wave_off = (((size_t)b * a.TB) * a.P + p0) * 8; // words
    const unsigned blk_stride = (unsigned)a.P * 32u;              // bytes per time block
    const unsigned extent = (unsigned)a.TB * blk_stride;
    auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.bq) + wave_off, 0, extent, 0x00020000);
    auto rout = __builtin_amdgcn_make_buffer_rsrc(a.xs + wave_off, 0, extent, 0x00020000);
    const unsigned voff = r < 2 ? (unsigned)(s * 32 + r * 16) : 0xFFFFFF00u;

    u32x4 ring[DEPTH];
    unsigned soff_ld = 0, soff_st = 0;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
        ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, soff_ld, 0);
        soff_ld += blk_stride; // the stream is padded to a multiple of DEPTH blocks
    }
    int32_t x = 0, tmp;
    for (int tb0 = 0; tb0 < a.TB; tb0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const u32x4 cur = ring[i];
            int32_t x1, x2, x3, x4;
            S5_SCAN_STEP("s_nop 1\n\t", "[2,3,0,1]", x, x1, cA, kA, cur[0]);
            S5_SCAN_STEP("", "[3,2,1,0]", x1, x2, cB, kB, cur[1]);
            S5_SCAN_STEP("", "[2,3,0,1]", x2, x3, cA, kA, cur[2]);
            S5_SCAN_STEP("", "[3,2,1,0]", x3, x4, cB, kB, cur[3]);
            x = x4;
            // refill this ring slot (clamped at the end of the stream; the extra data is never used)
            ring[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voff, soff_ld < extent ? soff_ld : extent - blk_stride, 0);
            soff_ld += blk_stride;
            u32x4 o;
            o[0] = (unsigned)x1; o[1] = (unsigned)x2; o[2] = (unsigned)x3; o[3] = (unsigned)x4;
            __builtin_amdgcn_raw_buffer_store_b128(o, rout, voff, soff_st, 0);
            soff_st += blk_stride;
        }
    }
}

#include "scan_quad_asm.inc"

// Hand-scheduled variant: the two wait states in front of every DPP add are filled with the loop's own
// buffer_load / buffer_store / s_add instructions (tools/gen_scan_asm.py).  Same algorithm, layout and
// exactness bound as k_scan_quad.  One wave per workgroup: every wave gets its own CU front end.
// The stream must be followed by S5_SCAN_ASM_DEPTH blocks of readable padding (the ring runs ahead).
__device__ __forceinline__ void scan_quad_group(ScanQuadArgs &a, const GroupOff &go)
{
    const int64_t g = blockIdx.y;
    gshift(a.bq, g * go.ws); gshift(a.xs, g * go.ws); gshift(a.run_if, g * go.ws);
    gshift(a.x0_re, g * go.state_in); gshift(a.x0_im, g * go.state_in);
}

__global__ __launch_bounds__(64) void k_scan_quad_asm(ScanQuadArgs a, GroupOff go)
{
    scan_quad_group(a, go);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int groups = a.P >> 4;
    const int b = wave / groups, p0 = (wave % groups) << 4;
    if (b >= a.B) return;
    const int s = lane >> 2, r = lane & 3;
    const int p = p0 + s;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int sre = 16 - a.ea_re, sim = 16 - a.ea_im;
    const int32_t kre = (1 << 16) - (1 << sre);
    int32_t cA, cB, kA = 0, kB = 0;
    if (r == 0) { cA = cB = Ar << sre; }
    else if (r == 1) { cA = cB = Ar << sim; }
    else if (r == 2) { cA = -(Ai << sre); kA = kre; cB = Ai << sim; }
    else { cA = Ai << sim; cB = -(Ai << sre); kB = kre; }
    // this launch covers time blocks [tb0, tb0 + ntb) (ntb % DEPTH == 0; ntb == 0: up to TB): a chunk of the
    // bproj | scan | cgate pipeline starts from the last state of the previous chunk.  After an odd step lanes
    // 0,3 of a quad hold the real part and lanes 1,2 the imaginary part.
    const int tb0 = a.tb0, ntb = a.ntb > 0 ? a.ntb : a.TB - a.tb0;
    int32_t x0 = quad_x0(a, b, p, r);
    if (tb0 > 0) x0 = a.xs[native_word(b, 4 * tb0 - 1, p, (r == 0 || r == 3) ? 0 : 1, a.TB, a.P)];
    const size_t wave_off = (((size_t)b * a.TB + tb0) * a.P + p0) * 8; // words
    const unsigned blk_stride = (unsigned)a.P * 32u;
    const unsigned extent = (unsigned)(a.TB - tb0) * blk_stride;
    const unsigned long long bin = (unsigned long long)(a.bq + wave_off), bout = (unsigned long long)(a.xs + wave_off);
    u32x4 rin, rout;
    rin[0] = (unsigned)bin; rin[1] = (unsigned)(bin >> 32) & 0xffffu; rin[2] = extent; rin[3] = 0x00020000u;
    rout[0] = (unsigned)bout; rout[1] = (unsigned)(bout >> 32) & 0xffffu; rout[2] = extent; rout[3] = 0x00020000u;
    const unsigned voff = r < 2 ? (unsigned)(s * 32 + r * 16) : 0xFFFFFF00u;
    unsigned sld = 0, sst = 0, cnt = (unsigned)ntb / S5_SCAN_ASM_DEPTH;
    asm volatile(S5_SCAN_ASM_BODY
                 : [sld] "+s"(sld), [sst] "+s"(sst), [cnt] "+s"(cnt)
                 : [ca] "v"(cA), [cb] "v"(cB), [ka] "v"(kA), [kb] "v"(kB), [voff] "v"(voff), [x0] "v"(x0), [rin] "s"(rin),
                   [rout] "s"(rout), [stride] "s"(blk_stride)
                 : S5_SCAN_ASM_CLOBBERS);
}

// int16 streams (S5FXP_FWD_DEFER_REDO forwards, where the caller checks the status words): the same chain; the
// input halfword is picked by the SDWA source select, the output is packed with saturation -- a state beyond 16 bits
// is stored as +-32767 / -32768, which the consumer's range check treats as out of range.  Half the bytes of
// k_scan_quad_asm on both sides; item = 8 bytes (4 steps of one component), state block = 16 bytes.
__global__ __launch_bounds__(64) void k_scan_quad_asm16(ScanQuadArgs a, GroupOff go)
{
    scan_quad_group(a, go);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int groups = a.P >> 4;
    const int b = wave / groups, p0 = (wave % groups) << 4;
    if (b >= a.B) return;
    const int s = lane >> 2, r = lane & 3;
    const int p = p0 + s;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int sre = 16 - a.ea_re, sim = 16 - a.ea_im;
    const int32_t kre = (1 << 16) - (1 << sre);
    int32_t cA, cB, kA = 0, kB = 0;
    if (r == 0) { cA = cB = Ar << sre; }
    else if (r == 1) { cA = cB = Ar << sim; }
    else if (r == 2) { cA = -(Ai << sre); kA = kre; cB = Ai << sim; }
    else { cA = Ai << sim; cB = -(Ai << sre); kB = kre; }
    const int32_t x0 = quad_x0(a, b, p, r);
    const size_t wave_off = (((size_t)b * a.TB) * a.P + p0) * 8; // halfwords
    const unsigned blk_stride = (unsigned)a.P * 16u;
    const unsigned extent = (unsigned)a.TB * blk_stride;
    const unsigned long long bin = (unsigned long long)(reinterpret_cast<const int16_t *>(a.bq) + wave_off),
                             bout = (unsigned long long)(reinterpret_cast<int16_t *>(a.xs) + wave_off);
    u32x4 rin, rout;
    rin[0] = (unsigned)bin; rin[1] = (unsigned)(bin >> 32) & 0xffffu; rin[2] = extent; rin[3] = 0x00020000u;
    rout[0] = (unsigned)bout; rout[1] = (unsigned)(bout >> 32) & 0xffffu; rout[2] = extent; rout[3] = 0x00020000u;
    // out-of-range buffer offsets: loads return 0, stores are dropped -- the lanes that do not touch memory, and every lane of a
    // padding slot of a compacted layer (its Bu is 0 and so is its state)
    const unsigned voff = r < 2 && (a.live_slots <= 0 || p < a.live_slots) ? (unsigned)(s * 16 + r * 8) : 0xFFFFFF00u;
    unsigned sld = 0, sst = 0, cnt = (unsigned)a.TB / S5_SCAN_ASM_DEPTH;
    asm volatile(S5_SCAN16_ASM_BODY
                 : [sld] "+s"(sld), [sst] "+s"(sst), [cnt] "+s"(cnt)
                 : [ca] "v"(cA), [cb] "v"(cB), [ka] "v"(kA), [kb] "v"(kB), [voff] "v"(voff), [x0] "v"(x0), [rin] "s"(rin),
                   [rout] "s"(rout), [stride] "s"(blk_stride)
                 : S5_SCAN16_ASM_CLOBBERS);
}

// The exact recurrence in the same quad layout, for states of any width (the re-run behind the range check and
// S5FXP_FWD_EXACT): the reference's int32 products (v_mul_lo_u32 wraps exactly like them), arithmetic shift, the
// subtraction as a conditional negation -(a >> e) == ((a >> e) ^ -1) + 1 folded into a three-operand add with Bu.
// Five dependent instructions per step instead of three, one of them quarter rate: ~50 cycles per step against the
// one-lane-per-state kernel's ~190 (k_scan_lane_native: 360 us per layer at B=32, L=4096).  int32 streams.
__global__ __launch_bounds__(64) void k_scan_quad32_asm(ScanQuadArgs a, GroupOff go)
{
    scan_quad_group(a, go);
    if (a.run_if && *a.run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int groups = a.P >> 4;
    const int b = wave / groups, p0 = (wave % groups) << 4;
    if (b >= a.B) return;
    const int s = lane >> 2, r = lane & 3;
    const int p = p0 + s;
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    // even steps (A): lanes 0,3 hold re, lanes 1,2 im; lane 0: Ar*re, lane 2: -(Ai*im) -> re'; lane 1: Ar*im,
    // lane 3: Ai*re -> im'.  Odd steps (B): lanes 0,2 hold re, 1,3 im; lane 3: -(Ai*im) -> re'; lane 2: Ai*re -> im'.
    int32_t cA, cB, sA, sB, mA = 0, mB = 0;
    if (r == 0) { cA = cB = Ar; sA = sB = a.ea_re; }
    else if (r == 1) { cA = cB = Ar; sA = sB = a.ea_im; }
    else if (r == 2) { cA = cB = Ai; sA = a.ea_re; mA = -1; sB = a.ea_im; }
    else { cA = cB = Ai; sA = a.ea_im; sB = a.ea_re; mB = -1; }
    const int32_t oA = mA & 1, oB = mB & 1, x0 = quad_x0(a, b, p, r);
    const size_t wave_off = (((size_t)b * a.TB) * a.P + p0) * 8; // words
    const unsigned blk_stride = (unsigned)a.P * 32u;
    const unsigned extent = (unsigned)a.TB * blk_stride;
    const unsigned long long bin = (unsigned long long)(a.bq + wave_off), bout = (unsigned long long)(a.xs + wave_off);
    u32x4 rin, rout;
    rin[0] = (unsigned)bin; rin[1] = (unsigned)(bin >> 32) & 0xffffu; rin[2] = extent; rin[3] = 0x00020000u;
    rout[0] = (unsigned)bout; rout[1] = (unsigned)(bout >> 32) & 0xffffu; rout[2] = extent; rout[3] = 0x00020000u;
    const unsigned voff = r < 2 ? (unsigned)(s * 32 + r * 16) : 0xFFFFFF00u;
    unsigned sld = 0, sst = 0, cnt = (unsigned)a.TB / S5_SCAN_ASM_DEPTH;
    asm volatile(S5_SCAN32W_ASM_BODY
                 : [sld] "+s"(sld), [sst] "+s"(sst), [cnt] "+s"(cnt)
                 : [ca] "v"(cA), [cb] "v"(cB), [sa] "v"(sA), [sb] "v"(sB), [ma] "v"(mA), [mb] "v"(mB), [oa] "v"(oA),
                   [ob] "v"(oB), [voff] "v"(voff), [x0] "v"(x0), [rin] "s"(rin), [rout] "s"(rout), [stride] "s"(blk_stride)
                 : S5_SCAN32W_ASM_CLOBBERS);
}

// ---------------------------------------------------------------------------------------------------------------
// Pair kernel: two lanes per state, four instructions per step (tools/gen_scan_asm.py, "Pair kernel"):
//     z1 = c_own * x + K               K = (Bu << 16) + k, written ready-made by the B projection (k_bproj_p<.., SM = 2>)
//     z2 = c_part * x[partner]         DPP read of the partner lane's state
//     x' = (z1 >> 16) + (z2 >> 16)
// The lane that holds re computes im' = asr(Ai*re) + asr(Ar*im[partner]) + Bu_im, the lane that holds im computes
// re' = -asr(Ai*im) + asr(Ar*re[partner]) + Bu_re: a lane's role alternates every step, both own products carry Ai
// (the negated one with k = 2^16 - 2^(16-e)), both partner products Ar.  Lane A (even) holds re before even steps,
// lane B im.
//
// Exactness: z2 < 2^31 needs |Ar| * 2^(16-e) * |x| < 2^31; z1 needs |Ai| * 2^(16-e) * |x| + 2^16 * (|Bu| + 1) < 2^31.
// With |Bu| <= bmax known statically (its bits minus the shift to the state exponent) that is a bound |x| <= xmax
// (pack_layer: LayerDev::pair_xmax), checked by the consumer on every stored state exactly like the quad kernels'
// bound; states are stored as saturated int16, so a state beyond 16 bits fails the check as well.
//
// Streams ("pair-native", a wave's blocks are contiguous so that immediate offsets reach eight of them):
//   K   int32  word(b, tb, p, lane, slot) = ((((b*PG + p/32)*TB + tb)*32 + p%32)*2 + lane)*4 + slot,  PG = P/32,
//              slots of a lane = steps [t0, t2, t1, t3] of block tb: lane A = [Kim0, Kim2, Kre1, Kre3],
//              lane B = [Kre0, Kre2, Kim1, Kim3]  (each producer lane of the B projection writes two 8-byte halves)
//   xs  int16  half(b, t8, p, lane, j)    = ((((b*PG + p/32)*TB/2 + t8)*32 + p%32)*2 + lane)*8 + j, 8 steps per item:
//              lane A = [im0 im2 | re1 re3 | im4 im6 | re5 re7], lane B = [re0 re2 | im1 im3 | re4 re6 | im5 im7]
__device__ __forceinline__ int64_t pair_word(int64_t b, int tb, int p, int TB, int P)
{
    return ((((b * (P >> 5) + (p >> 5)) * TB + tb) << 5) + (p & 31)) << 3;
}

// The state after step L-1 of every (sequence, state), read back from the stream the recurrence wrote (the streaming
// carry out).  mode 0: scan-native int32, 1: scan-native int16, 2: pair-native int16.  With redo != nullptr and *redo
// != 0 the exact re-run has rewritten the stream as scan-native int32 (mode 0) whatever the fast kernel's mode was.
__global__ void k_state_out(const void *xs, int mode, const int32_t *redo, int B, int L, int P, int TB, int32_t *out_re,
                            int32_t *out_im, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(xs, g * go.ws); gshift(redo, g * go.ws); gshift(out_re, g * go.state_out); gshift(out_im, g * go.state_out);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * P) return;
    const int b = i / P, p = i % P, t = L - 1;
    if (redo && *redo) mode = 0;
    int32_t re, im;
    if (mode == 0) {
        const int32_t *x = reinterpret_cast<const int32_t *>(xs);
        re = x[native_word(b, t, p, 0, TB, P)];
        im = x[native_word(b, t, p, 1, TB, P)];
    } else if (mode == 1) {
        const int16_t *x = reinterpret_cast<const int16_t *>(xs);
        re = x[native_word(b, t, p, 0, TB, P)];
        im = x[native_word(b, t, p, 1, TB, P)];
    } else {
        const int16_t *x = reinterpret_cast<const int16_t *>(xs) + (pair_word(b, t >> 3, p, TB >> 1, P) << 1);
        const int j = t & 7, jj = j & 3;
        const int hw = (2 * (j >> 2) + (jj & 1)) * 2 + (jj >> 1); // position of step j in a lane's 8-step item
        const int16_t va = x[hw], vb = x[8 + hw];                  // lane A, lane B
        re = (j & 1) ? va : vb;                                    // even steps: lane A computed im, lane B re
        im = (j & 1) ? vb : va;
    }
    out_re[i] = re;
    out_im[i] = im;
}

struct ScanPairArgs {
    const int32_t *k;           // pair-native K stream
    int16_t *xs;                // pair-native packed states
    const int32_t *a_re, *a_im; // (P)
    int32_t B, TB, P;           // TB % S5_SCANP_ASM_DEPTH == 0
    int32_t ea_re, ea_im;
    const int32_t *x0_re, *x0_im; // (B,P) state before the first step (streaming carry), nullptr = zeros
};

__global__ __launch_bounds__(64) void k_scan_pair_asm(ScanPairArgs a, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(a.k, g * go.ws); gshift(a.xs, g * go.ws); gshift(a.x0_re, g * go.state_in); gshift(a.x0_im, g * go.state_in);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x); // (b, state group of 32)
    if (wave >= a.B * (a.P >> 5)) return;
    const int p = ((wave % (a.P >> 5)) << 5) + (lane >> 1);
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int sre = 16 - a.ea_re, sim = 16 - a.ea_im;
    const bool laneB = lane & 1;
    // even steps: lane A holds re -> computes im', lane B holds im -> computes re'; odd steps the other way round
    const int32_t c_im_own = Ai << sim, c_im_part = Ar << sim, c_re_own = -(Ai << sre), c_re_part = Ar << sre;
    const int32_t coe = laneB ? c_re_own : c_im_own, cpe = laneB ? c_re_part : c_im_part;
    const int32_t coo = laneB ? c_im_own : c_re_own, cpo = laneB ? c_im_part : c_re_part;
    const unsigned long long pin = (unsigned long long)(a.k + (size_t)wave * a.TB * 256),
                             pout = (unsigned long long)(a.xs + (size_t)wave * a.TB * 256);
    const unsigned vin = lane * 16 + 4096, vout = vin;
    // lane A holds re before an even step, lane B im
    const size_t sp = (size_t)(wave / (a.P >> 5)) * a.P + p;
    const int32_t x0 = a.x0_re ? (laneB ? a.x0_im[sp] : a.x0_re[sp]) : 0;
    unsigned cnt = (unsigned)a.TB / S5_SCANP_ASM_DEPTH;
    asm volatile(S5_SCANP_ASM_BODY
                 : [cnt] "+s"(cnt)
                 : [coe] "v"(coe), [cpe] "v"(cpe), [coo] "v"(coo), [cpo] "v"(cpo), [vin] "v"(vin), [vout] "v"(vout),
                   [x0] "v"(x0), [pin] "s"(pin), [pout] "s"(pout)
                 : S5_SCANP_ASM_CLOBBERS);
}

// ---------------------------------------------------------------------------------------------------------------
// The same pair recurrence fed from LDS: workgroup = the computing wave + a HELPER wave on another SIMD of the CU.
// The helper streams the layer's Bu as int16 (half the bytes of the int32 K stream), expands K = (Bu << 16) + k and
// writes it where the computing wave's ds_read_b128 expects it; three LDS buffers of BLOCKS time blocks rotate,
// one s_barrier per buffer.  The computing wave's loop is tools/gen_scan_asm.py "pairl".
//
// Bu stream ("pair16-native", written by k_bproj_p<.., SM = 3>): per wave run (b, state group of 32), per PAIR of
// blocks, per lane l = 2 * (p % 32) + {A, B}: 8 halfwords = [block 2j: t0 t2 t1 t3 | block 2j+1: t0 t2 t1 t3] -- one
// coalesced 1 KB line per block pair, 16 bytes per helper lane.
__device__ __forceinline__ int64_t pair16_half(int64_t b, int tb, int p, int lane_sel, int TB, int P)
{
    return ((((b * (P >> 5) + (p >> 5)) * (TB >> 1) + (tb >> 1)) << 6) + 2 * (p & 31) + lane_sel) * 8 + 4 * (tb & 1);
}

struct ScanPairLArgs {
    const int16_t *b16;         // pair16-native Bu stream (already shifted to the state exponent)
    int16_t *xs;                // pair-native packed states
    const int32_t *a_re, *a_im; // (P)
    int32_t B, TB, P;           // TB % BLOCKS == 0 (k_scan_pairl_asm<BLOCKS>)
    int32_t ea_re, ea_im;
    const int32_t *x0_re, *x0_im; // (B,P) state before the first step (streaming carry), nullptr = zeros
    // lanes 2p, 2p+1 belong to state slot p of the wave's 32.  live_slots > 0 (a compacted layer, s5fxp_fast.hpp: the live
    // states, rounded up to an even number so that whole lane quads remain): the lanes of the slots at or above it leave at
    // once in BOTH waves, so the padding slots' share of the two streams is neither read nor written (their producer and
    // consumer skip the same slots: k_bproj_p<.., SM = 3> and k_cgate_p<.., PAIR>)
    int32_t live_slots;
};

template <int... J, class F>
__device__ __forceinline__ void for_items(std::integer_sequence<int, J...>, F f)
{
    (f(std::integral_constant<int, J>{}), ...);
}

// DBG (tools/ubench_pair.hip only): 1 = the helper skips its loads, 2 = it only meets the barriers
template <int BLOCKS, int DBG = 0> // time blocks per LDS buffer (16 or 32): one s_barrier per BLOCKS * 4 steps; 3 * BLOCKS KB of dynamic LDS
__global__ __launch_bounds__(128) void k_scan_pairl_asm(ScanPairLArgs a, GroupOff go)
{
    {
        const int64_t g = blockIdx.y;
        gshift(a.b16, g * go.ws); gshift(a.xs, g * go.ws); gshift(a.x0_re, g * go.state_in); gshift(a.x0_im, g * go.state_in);
    }
    extern __shared__ __attribute__((aligned(16))) int32_t kbuf[]; // three buffers of BLOCKS KB
    constexpr int BUFW = BLOCKS * 256;                              // words per buffer
    const int lane = threadIdx.x & 63;
    if (a.live_slots > 0 && (((int)blockIdx.x % (a.P >> 5)) << 5) + (lane >> 1) >= a.live_slots) return; // padding slots of a compacted layer
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // 0: recurrence, 1: helper
    const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x);         // (b, state group of 32)
    const int n_it = a.TB / BLOCKS;
    const int sre = 16 - a.ea_re;
    const bool laneB = lane & 1;
    if (role >= 1) {
        // ---- helper: iteration k+2 goes into buffer (k+2) % 3 while the recurrence works on k
        constexpr int ITEMS = BLOCKS / 2, NSETS = 32 / ITEMS; // 16-byte items per lane and iteration; register sets
        static_assert(ITEMS == 8 || ITEMS == 16, "the asm below moves eight 16-byte items per lane at a time");
        const i32x4 *src = reinterpret_cast<const i32x4 *>(a.b16) + (size_t)wave * (a.TB >> 1) * 64 + lane;
        const int32_t k_re = 65536 - (1 << sre);
        const int32_t kE = laneB ? k_re : 0, kO = laneB ? 0 : k_re; // even steps: lane B computes re' (which carries k)
        // 32 items (128 registers) of loads are in flight, as NSETS sets: the loads of an iteration are issued ~256 steps
        // (~2.5 us of recurrence) before they are used.  The loads and their waits are spelled in asm: across this
        // loop's back edge the compiler's vmcnt bookkeeping falls back to "wait for everything", which puts a whole HBM
        // latency into every round (ubench_pair: 59 us instead of 40).  Loads return in order, every round issues
        // exactly ITEMS loads after waiting for the oldest set, so "all but the youngest (NSETS - 1) * ITEMS" is
        // precisely that set.  Past the end the last iteration is re-read and expanded into a buffer nobody reads any
        // more: no conditional memory instruction anywhere.
        i32x4 r[NSETS][ITEMS];
        auto fetch8 = [&](i32x4 *q, const i32x4 *ptr) { // eight items, +-4 KB immediate offsets around ptr
            if constexpr (DBG == 0)
                asm volatile("global_load_dwordx4 %0, %8, off offset:-4096\n\tglobal_load_dwordx4 %1, %8, off offset:-3072\n\t"
                             "global_load_dwordx4 %2, %8, off offset:-2048\n\tglobal_load_dwordx4 %3, %8, off offset:-1024\n\t"
                             "global_load_dwordx4 %4, %8, off\n\tglobal_load_dwordx4 %5, %8, off offset:1024\n\t"
                             "global_load_dwordx4 %6, %8, off offset:2048\n\tglobal_load_dwordx4 %7, %8, off offset:3072"
                             : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
                             : "v"(ptr)
                             : "memory");
        };
        auto fetch = [&](i32x4(&q)[ITEMS], int it) {
            it = it < n_it ? it : n_it - 1;
            const i32x4 *ptr = src + ((size_t)it * ITEMS + 4) * 64;
            fetch8(q, ptr);
            if constexpr (ITEMS == 16) fetch8(q + 8, ptr + 8 * 64);
        };
        auto ready = [&](i32x4(&q)[ITEMS]) { // the oldest set has landed
            if constexpr (ITEMS == 8)
                asm volatile("s_waitcnt vmcnt(24)"
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                             :
                             : "memory");
            else {
                asm volatile("s_waitcnt vmcnt(16)"
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
                             :
                             : "memory");
                asm volatile(""
                             : "+v"(q[8]), "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15])
                             :
                             : "memory");
            }
        };
        auto expand = [&](const i32x4(&q)[ITEMS], int it) {
            if constexpr (DBG == 2) return;
            int32_t *dst = kbuf + (it % 3) * BUFW + lane * 4;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) { // [t0 t2 | t1 t3] halfword pairs of block 2j + h
                    const uint32_t d0 = (uint32_t)q[j][2 * h], d1 = (uint32_t)q[j][2 * h + 1];
                    i32x4 k;
                    k[0] = (int)((d0 << 16) + (uint32_t)kE);
                    k[1] = (int)((d0 & 0xffff0000u) | (uint32_t)kE);
                    k[2] = (int)((d1 << 16) + (uint32_t)kO);
                    k[3] = (int)((d1 & 0xffff0000u) | (uint32_t)kO);
                    *reinterpret_cast<i32x4 *>(dst + (2 * j + h) * 256) = k;
                }
            }
        };
        auto meet = [&]() { // the LDS writes have landed (the clobber keeps the compiler from sinking them below the barrier)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) r[s][j] = i32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < NSETS; ++s) fetch(r[s], s);
        ready(r[0]); expand(r[0], 0); fetch(r[0], NSETS);
        ready(r[1]); expand(r[1], 1); fetch(r[1], NSETS + 1);
        meet();
        // round k (the recurrence works on iteration k): iteration k+2 goes into LDS from set (k+2) % NSETS, which then
        // receives iteration k+2+NSETS
        // One round, item by item: expand item j of the set that has landed (two ds_write_b128), refill its registers with
        // item j of the iteration NSETS rounds ahead.  (Tried and measured in tools/ubench_pair, all within 0.2 us of this:
        // bursts instead of item-by-item; an s_sleep after every item -- 2 x 64 clocks already starves the recurrence;
        // the helper as wave 2 or 3 of the workgroup -- wave 2 shares the computing wave's LDS path: +1.7 us; extra waves
        // that share the fill of the first two buffers.  What this wave costs the computing one, +3 us over a helper that
        // only meets the barriers, is the LDS writes and the loads themselves, half each.)
        auto round = [&](i32x4(&q)[ITEMS], int kk) {
            ready(q);
            int itf = kk + 2 + NSETS;
            itf = itf < n_it ? itf : n_it - 1;
            const i32x4 *ptr = src + ((size_t)itf * ITEMS + 4) * 64;
            int32_t *dst = kbuf + ((kk + 2) % 3) * BUFW + lane * 4;
            for_items(std::make_integer_sequence<int, ITEMS>{}, [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (DBG != 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t d0 = (uint32_t)q[j][2 * h], d1 = (uint32_t)q[j][2 * h + 1];
                        i32x4 k;
                        k[0] = (int)((d0 << 16) + (uint32_t)kE);
                        k[1] = (int)((d0 & 0xffff0000u) | (uint32_t)kE);
                        k[2] = (int)((d1 << 16) + (uint32_t)kO);
                        k[3] = (int)((d1 & 0xffff0000u) | (uint32_t)kO);
                        *reinterpret_cast<i32x4 *>(dst + (2 * j + h) * 256) = k;
                    }
                }
                const i32x4 *pj = ptr + (j / 8) * 8 * 64; // +-4 KB immediate offsets around it
                if constexpr (DBG == 0)
                    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(q[j]) : "v"(pj), "n"((j % 8 - 4) * 1024) : "memory");
            });
            meet();
        };
        for (int k = 0; k < n_it; k += NSETS) {
            if constexpr (NSETS == 4) {
                round(r[2], k);
                if (k + 1 >= n_it) break;
                round(r[3], k + 1);
                if (k + 2 >= n_it) break;
                round(r[0], k + 2);
                if (k + 3 >= n_it) break;
                round(r[1], k + 3);
            } else {
                round(r[0], k);
                if (k + 1 >= n_it) break;
                round(r[1], k + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // nothing may still be landing in registers when the wave ends
        return;
    }
    const int sim = 16 - a.ea_im;
    const int p = ((wave % (a.P >> 5)) << 5) + (lane >> 1);
    const int32_t Ar = a.a_re[p], Ai = a.a_im[p];
    const int32_t c_im_own = Ai << sim, c_im_part = Ar << sim, c_re_own = -(Ai << sre), c_re_part = Ar << sre;
    const int32_t coe = laneB ? c_re_own : c_im_own, cpe = laneB ? c_re_part : c_im_part;
    const int32_t coo = laneB ? c_im_own : c_re_own, cpo = laneB ? c_im_part : c_re_part;
    const unsigned long long pout = (unsigned long long)(a.xs + (size_t)wave * a.TB * 256);
    const unsigned vlds = (unsigned)(size_t)kbuf + lane * 16, vout = lane * 16 + 4096;
    const size_t sp = (size_t)(wave / (a.P >> 5)) * a.P + p;
    const int32_t x0 = a.x0_re ? (laneB ? a.x0_im[sp] : a.x0_re[sp]) : 0; // lane A holds re before an even step

    unsigned cnt = (unsigned)n_it;
    if constexpr (BLOCKS == 16)
        asm volatile(S5_SCANPL16_ASM_BODY
                     : [cnt] "+s"(cnt)
                     : [coe] "v"(coe), [cpe] "v"(cpe), [coo] "v"(coo), [cpo] "v"(cpo), [vlds] "v"(vlds), [vout] "v"(vout),
                       [x0] "v"(x0), [pout] "s"(pout)
                     : S5_SCANPL16_ASM_CLOBBERS);
    else
        asm volatile(S5_SCANPL32_ASM_BODY
                     : [cnt] "+s"(cnt)
                     : [coe] "v"(coe), [cpe] "v"(cpe), [coo] "v"(coo), [cpo] "v"(cpo), [vlds] "v"(vlds), [vout] "v"(vout),
                       [x0] "v"(x0), [pout] "s"(pout)
                     : S5_SCANPL32_ASM_CLOBBERS);
}

} // namespace s5
