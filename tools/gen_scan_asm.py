#!/usr/bin/env python3
"""Generates sparsernns_amd/csrc/scan_quad_asm.inc: the hand-scheduled main loops of the quad
recurrence kernels (see scan_quad.hpp for the algorithm).

Why hand-scheduled: on gfx950 a DPP source written by the preceding VALU instruction needs two wait
states, and for a lone wave every wait state is a 4-cycle issue slot.  The chain per step is
    v_mad_i32_i24 -> v_add_u32 (SDWA) -> [2 wait states] -> v_add_u32 (DPP)
so the two slots are filled with the loop's own buffer_load / buffer_store / s_add / s_waitcnt
instead of s_nop (tools/ubench_scan.hip measures 26.8 cycles/step with nops, 17.0 for the bare
chain; the fillers make the memory traffic free).

Two bodies:
  S5_SCAN_ASM_BODY    int32 streams: 16-byte items (4 steps of one component) in and out
  S5_SCAN32W_ASM_BODY the exact chain for states of any width (the re-run behind the range check): 32-bit
                      v_mul_lo_u32, arithmetic shift, conditional negation, v_add3; int32 streams
  S5_SCAN16_ASM_BODY  int16 streams: 8-byte items; the input word is picked by the SDWA source select
                      (sext WORD_0 / WORD_1) at no cost, the four new states of a block are packed by two
                      v_cvt_pk_i16_i32 (saturating: a state beyond 16 bits comes out as +-32767/8, which the
                      consumer treats as "out of range" -> exact re-run) in two of the otherwise idle filler slots

Register plan (physical, clobbered):  ring slot i = v[R0+W*i : R0+W*i+W-1] (W = 4 or 2), output tuples A/B,
packed pair (16-bit variant), one temp.
Run:  python tools/gen_scan_asm.py
"""
import os

DEPTH = 16
PERM = ("[2,3,0,1]", "[3,2,1,0]")  # phase A (even steps), phase B (odd steps)


class Plan:
    def __init__(self, s16: bool, wide: bool = False):
        self.s16 = s16
        self.wide = wide                        # exact 32-bit chain (any state width)
        self.W = 2 if s16 else 4               # ring registers per block
        self.R0 = 32
        self.OA = self.R0 + self.W * DEPTH      # output tuple A (4 regs)
        self.OB = self.OA + 4
        self.PK = self.OB + 4                   # packed int16 pairs (16-bit variant)
        self.TMP = self.PK + 2
        self.LAST = self.TMP
        self.ld = "buffer_load_dwordx2" if s16 else "buffer_load_dwordx4"
        self.st = "buffer_store_dwordx2" if s16 else "buffer_store_dwordx4"

    def ring(self, i):
        return f"v[{self.R0 + self.W * i}:{self.R0 + self.W * i + self.W - 1}]"


def step(P, c, k, xprev, i, j, out_reg, perm, f1, f2):
    if P.wide:
        # (A*x wraps like the reference's int32 product) >> e, negated on the lane that owes -floor(.): -(a >> e) ==
        # ((a >> e) ^ -1) + 1, so one xor with a per-lane mask and a three-operand add that brings Bu and the +1
        ph = "a" if j % 2 == 0 else "b"
        return [
            f"v_mul_lo_u32 v{P.TMP}, {c}, v{xprev}",
            f"v_ashrrev_i32 v{P.TMP}, %[s{ph}], v{P.TMP}",
            f"v_xor_b32 v{P.TMP}, %[m{ph}], v{P.TMP}",
            f"v_add3_u32 v{P.TMP}, v{P.TMP}, v{P.R0 + 4 * i + j}, %[o{ph}]",
            f1, f2,
            f"v_add_u32_dpp v{out_reg}, v{P.TMP}, v{P.TMP} quad_perm:{perm} row_mask:0xf bank_mask:0xf",
        ]
    if P.s16:
        src1, sel = f"sext(v{P.R0 + 2 * i + (j >> 1)})", f"WORD_{j & 1}"
    else:
        src1, sel = f"v{P.R0 + 4 * i + j}", "DWORD"
    return [
        f"v_mad_i32_i24 v{P.TMP}, {c}, v{xprev}, {k}",
        f"v_add_u32_sdwa v{P.TMP}, sext(v{P.TMP}), {src1} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:{sel}",
        f1, f2,
        f"v_add_u32_dpp v{out_reg}, v{P.TMP}, v{P.TMP} quad_perm:{perm} row_mask:0xf bank_mask:0xf",
    ]


def block(P, i, first_of_kernel=False):
    """Block i of an iteration: consumes ring slot i, writes tuple (i & 1); its filler slots refill ring
    slot i-1 and store the previous block's tuple."""
    out = P.OA if (i & 1) == 0 else P.OB
    prev = P.OB if (i & 1) == 0 else P.OA
    prev_slot = (i - 1) % DEPTH
    nop = "s_nop 0"
    wait = f"s_waitcnt vmcnt({2 * DEPTH - 3})"
    if first_of_kernel:
        fill = [nop] * 8
    elif P.s16:
        fill = [
            f"v_cvt_pk_i16_i32 v{P.PK}, v{prev}, v{prev + 1}",
            f"v_cvt_pk_i16_i32 v{P.PK + 1}, v{prev + 2}, v{prev + 3}",
            f"{P.ld} {P.ring(prev_slot)}, %[voff], %[rin], %[sld] offen",
            "s_add_u32 %[sld], %[sld], %[stride]",
            f"{P.st} v[{P.PK}:{P.PK + 1}], %[voff], %[rout], %[sst] offen",
            "s_add_u32 %[sst], %[sst], %[stride]",
            nop, nop,
        ]
    else:
        fill = [
            f"{P.ld} {P.ring(prev_slot)}, %[voff], %[rin], %[sld] offen",
            "s_add_u32 %[sld], %[sld], %[stride]",
            f"{P.st} v[{prev}:{prev + 3}], %[voff], %[rout], %[sst] offen",
            "s_add_u32 %[sst], %[sst], %[stride]",
            nop, nop, nop, nop,
        ]
    lines = [wait]
    xprev = prev + 3
    for j in range(4):
        c, k = ("%[ca]", "%[ka]") if j % 2 == 0 else ("%[cb]", "%[kb]")
        lines += step(P, c, k, xprev, i, j, out + j, PERM[j % 2], fill[2 * j], fill[2 * j + 1])
        xprev = out + j
    return lines


def body(P):
    b = []
    # the state before the first step (0, or the last state of the previous chunk): the "previous" tuple of
    # block 0 is B
    b.append(f"v_mov_b32 v{P.OB + 3}, %[x0]")
    # a lone latency-bound wave: when projection kernels share the SIMD it must win every issue arbitration
    b.append("s_setprio 3")
    # prologue: fill the ring, then wait for all of it (once per kernel, ~1 memory latency)
    for i in range(DEPTH):
        b.append(f"{P.ld} {P.ring(i)}, %[voff], %[rin], %[sld] offen")
        b.append("s_add_u32 %[sld], %[sld], %[stride]")
    b.append("s_waitcnt vmcnt(0)")
    # peeled block 0 of the first iteration (nothing to store / refill yet), then blocks 1..DEPTH-1
    b += block(P, 0, first_of_kernel=True)
    for i in range(1, DEPTH):
        b += block(P, i)
    b.append("s_sub_u32 %[cnt], %[cnt], 1")
    b.append("s_cmp_eq_u32 %[cnt], 0")
    b.append("s_cbranch_scc1 2f")
    b.append("1:")
    for i in range(DEPTH):
        b += block(P, i)
    b.append("s_sub_u32 %[cnt], %[cnt], 1")
    b.append("s_cmp_lg_u32 %[cnt], 0")
    b.append("s_cbranch_scc1 1b")
    b.append("2:")
    # epilogue: the last block (index DEPTH-1, tuple B) has not been stored yet
    last = P.OB if ((DEPTH - 1) & 1) else P.OA
    if P.s16:
        b.append(f"v_cvt_pk_i16_i32 v{P.PK}, v{last}, v{last + 1}")
        b.append(f"v_cvt_pk_i16_i32 v{P.PK + 1}, v{last + 2}, v{last + 3}")
        b.append("s_nop 1")
        b.append(f"{P.st} v[{P.PK}:{P.PK + 1}], %[voff], %[rout], %[sst] offen")
    else:
        b.append("s_nop 1")
        b.append(f"{P.st} v[{last}:{last + 3}], %[voff], %[rout], %[sst] offen")
    b.append("s_waitcnt vmcnt(0)")
    return b


# ---------------------------------------------------------------------------------------------------------
# Pair kernel (round 2): TWO lanes per state, Bu folded into the multiply's addend, cross-lane read at the INPUT
# of the step.  tools/ubench_chain (profiles/r02_ubench_chain*.log) measured, for a lone wave: a step costs
# 4 cycles per instruction + 2, a memory instruction in a filler slot costs 6 (global_*) to 9 (buffer_*) cycles
# more than a nop.  Round 1's step is 5 instructions (mad, sdwa-add, 2 fillers for the DPP hazard, dpp-add); this
# one is 4:
#     F                                   filler slot: the two wait states a DPP read of x needs after its write
#     z1 = c_own * x + K                  v_mad_i32_i24; K = (Bu << 16) + k comes ready-made from the B projection
#     z2 = c_part * x[partner]            v_mul_i32_i24_dpp quad_perm:[1,0,3,2]
#     x' = (z1 >> 16) + (z2 >> 16)        v_add_u32_sdwa, both sources sign-extended WORD_1
# The lane that holds re computes im' (own product Ai*re, partner product Ar*im) and vice versa, so the roles of a
# lane alternate every step and both own products carry Ai (the negated one with its k) -- see scan_quad.hpp.
# Memory: one global_load_dwordx4 per 4 steps (K for 4 steps), one global_store_dwordx4 per 8 steps (8 states as
# saturated int16), immediate offsets inside 8 KB windows.  DEPTH blocks of 4 steps are kept in flight.
PAIR_DEPTH = 32
PAIR_SLOT = (0, 2, 1, 3)  # ring register used by step 0..3 of a block: items are [t0, t2, t1, t3]


class PairPlan:
    R0 = 32
    O = R0 + 4 * PAIR_DEPTH       # two output tuples of 8 (one 8-step group each)
    PK = O + 16                   # two packed tuples of 4
    Z1 = PK + 8
    Z2 = Z1 + 1
    VIN = Z2 + 1                  # running byte offsets (per lane) of the load / store windows
    VOUT = VIN + 1
    SCR = VOUT + 1                # target of the peeled group's dummy load
    LAST = SCR


def pair_iteration(first: bool):
    """One loop iteration = PAIR_DEPTH blocks.  Returns a list of (text, kind) with kind in
    {'valu','load','store','wait','other'}; waits carry the ring slots they must cover."""
    Q = PairPlan
    D = PAIR_DEPTH
    out = []
    for i in range(D):
        g, half = i // 2, i % 2
        cur, prev = g & 1, (g & 1) ^ 1
        oc, op = Q.O + 8 * cur, Q.O + 8 * prev
        pk = Q.PK + 4 * prev          # the tuple that receives / stores the PREVIOUS group
        refill = (i - 1) % D          # ring slot whose block was consumed last
        n_blk = refill                # position of that slot's next block in the load sequence (mod 8 window)
        ld = (f"global_load_dwordx4 v[{Q.R0 + 4 * refill}:{Q.R0 + 4 * refill + 3}], v{Q.VIN}, %[pin] "
              f"offset:{((n_blk % 8) - 4) * 1024}")
        dead = first and g == 0       # no previous group yet
        nop = ("s_nop 0", "other")
        bump_in = n_blk % 8 == 7
        if first and i == 0:
            # slot D-1 still holds block D-1 of the prologue: nothing to refill yet (a dummy load keeps the count of
            # memory instructions in flight equal to the steady state's, which the vmcnt values assume)
            ld, bump_in = f"global_load_dword v{Q.SCR}, v{Q.VIN}, %[pin] offset:-4096", False
        if half == 0:
            fills = [(ld, "load"),
                     nop if dead else (f"v_cvt_pk_i16_i32 v{pk}, v{op}, v{op + 2}", "valu"),
                     nop if dead else (f"v_cvt_pk_i16_i32 v{pk + 1}, v{op + 1}, v{op + 3}", "valu"),
                     nop if dead else (f"v_cvt_pk_i16_i32 v{pk + 2}, v{op + 4}, v{op + 6}", "valu")]
        else:
            gp = (g - 1) % (D // 2)   # index of the previous group in the store sequence
            st = f"global_store_dwordx4 v{Q.VOUT}, v[{pk}:{pk + 3}], %[pout] offset:{((gp % 8) - 4) * 1024}"
            fills = [(ld, "load"),
                     nop if dead else (f"v_cvt_pk_i16_i32 v{pk + 3}, v{op + 5}, v{op + 7}", "valu"),
                     # the peeled first group keeps the memory-instruction count of the steady state (vmcnt bookkeeping)
                     (f"global_load_dword v{Q.SCR}, v{Q.VIN}, %[pin] offset:-4096", "store") if dead else (st, "store"),
                     (f"WAIT {i}", "wait") if i % 4 == 3 else nop]
        for s_ in range(4):
            j = 4 * half + s_
            xprev = oc + j - 1 if j > 0 else op + 7
            par = "e" if j % 2 == 0 else "o"
            out.append(fills[s_])
            if fills[s_][1] == "load" and bump_in:
                out.append((f"v_add_u32 v{Q.VIN}, 0x2000, v{Q.VIN}", "valu"))
            if fills[s_][1] == "store" and not dead and gp % 8 == 7:
                out.append((f"v_add_u32 v{Q.VOUT}, 0x2000, v{Q.VOUT}", "valu"))
            k = Q.R0 + 4 * i + PAIR_SLOT[s_]
            out.append((f"v_mad_i32_i24 v{Q.Z1}, %[co{par}], v{xprev}, v{k}", "valu"))
            out.append((f"v_mul_i32_i24_dpp v{Q.Z2}, v{xprev}, %[cp{par}] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "valu"))
            out.append((f"v_add_u32_sdwa v{oc + j}, sext(v{Q.Z1}), sext(v{Q.Z2}) dst_sel:DWORD dst_unused:UNUSED_PAD "
                        f"src0_sel:WORD_1 src1_sel:WORD_1", "valu"))
    return out


def pair_resolve_waits(it, like=None):
    """`WAIT i` sits in block i (i % 4 == 3) and must cover ring slots i+1 .. i+4 (the next four blocks; slot D is slot 0
    of the next iteration).  Slot j is refilled in block j+1, so the data of a slot whose refill comes LATER in the
    iteration than the wait was loaded one iteration earlier.  vmcnt = memory instructions issued after the youngest of
    those loads, minus one: the peeled first iteration replaces group 0's store by a dummy load to keep the count, the
    margin covers the prologue's different interleaving."""
    D = PAIR_DEPTH
    if like is not None:  # the peeled iteration: same positions, same values as the steady state
        skel = lambda seq: [k for _, k in seq if k in ("load", "store", "wait")]
        assert skel(it) == skel(like)  # same memory / wait skeleton
        vals = iter([t for t in pair_resolve_waits(like) if t.startswith("s_waitcnt")])
        return [next(vals) if k == "wait" else t for t, k in it]
    mem = [idx for idx, (t, k) in enumerate(it) if k in ("load", "store")]
    load_pos = {}
    for idx, (t, k) in enumerate(it):
        if k == "load":
            load_pos[(int(t.split("v[")[1].split(":")[0]) - PairPlan.R0) // 4] = idx
    res = []
    for idx, (t, k) in enumerate(it):
        if k != "wait":
            res.append(t)
            continue
        i = int(t.split()[1])
        worst = None
        for slot in range(i + 1, i + 5):
            p = load_pos[slot % D]
            if p > idx:   # issued in the previous iteration
                n = sum(1 for m in mem if m > p) + sum(1 for m in mem if m < idx)
            else:         # issued earlier in this iteration (the wrap: slots of the next iteration's first blocks)
                n = sum(1 for m in mem if p < m < idx)
            worst = n if worst is None else min(worst, n)
        assert 0 <= worst - 1 <= 63, worst
        res.append(f"s_waitcnt vmcnt({worst - 1})")
    return res


def pair_body():
    Q = PairPlan
    D = PAIR_DEPTH
    b = []
    b.append(f"v_mov_b32 v{Q.O + 15}, %[x0]")       # "previous" tuple of group 0 is tuple 1: its last register is x_(-1)
    b.append(f"v_mov_b32 v{Q.VIN}, %[vin]")
    b.append(f"v_mov_b32 v{Q.VOUT}, %[vout]")
    b.append("s_setprio 3")
    for n in range(D):                               # prologue: blocks 0 .. D-1
        b.append(f"global_load_dwordx4 v[{Q.R0 + 4 * n}:{Q.R0 + 4 * n + 3}], v{Q.VIN}, %[pin] offset:{((n % 8) - 4) * 1024}")
        if n % 8 == 7:
            b.append(f"v_add_u32 v{Q.VIN}, 0x2000, v{Q.VIN}")
    b.append("s_waitcnt vmcnt(0)")
    b += pair_resolve_waits(pair_iteration(True), like=pair_iteration(False))
    b.append("s_sub_u32 %[cnt], %[cnt], 1")
    b.append("s_cmp_eq_u32 %[cnt], 0")
    b.append("s_cbranch_scc1 2f")
    b.append("1:")
    b += pair_resolve_waits(pair_iteration(False))
    b.append("s_sub_u32 %[cnt], %[cnt], 1")
    b.append("s_cmp_lg_u32 %[cnt], 0")
    b.append("s_cbranch_scc1 1b")
    b.append("2:")
    last_g = D // 2 - 1
    o, pk = Q.O + 8 * (last_g & 1), Q.PK + 4 * (last_g & 1)
    b.append(f"v_cvt_pk_i16_i32 v{pk}, v{o}, v{o + 2}")
    b.append(f"v_cvt_pk_i16_i32 v{pk + 1}, v{o + 1}, v{o + 3}")
    b.append(f"v_cvt_pk_i16_i32 v{pk + 2}, v{o + 4}, v{o + 6}")
    b.append(f"v_cvt_pk_i16_i32 v{pk + 3}, v{o + 5}, v{o + 7}")
    b.append("s_nop 1")
    b.append(f"global_store_dwordx4 v{Q.VOUT}, v[{pk}:{pk + 3}], %[pout] offset:{((last_g % 8) - 4) * 1024}")
    b.append("s_waitcnt vmcnt(0)")
    return b


def emit_pair(name="S5_SCANP_ASM"):
    b = pair_body()
    clobbers = ", ".join(f'"v{r}"' for r in range(PairPlan.R0, PairPlan.LAST + 1))
    text = "\n".join(f'    "{l}\\n\\t"' for l in b)
    return (f"#define {name}_BODY \\\n{text.replace(chr(10), ' ' + chr(92) + chr(10))}\n"
            f'#define {name}_CLOBBERS {clobbers}, "memory", "scc"\n'), len(b)


# ---------------------------------------------------------------------------------------------------------
# Pair kernel fed from LDS ("pairl"): the same step, but K comes from a ring of three LDS buffers that a HELPER
# wave of the same workgroup fills from an int16 Bu stream (scan_quad.hpp k_scan_pairl_asm): the stream in HBM stays
# 16 bit (half the bytes of the int32 K stream), the expansion K = (Bu << 16) + k costs the helper's issue slots, not
# this wave's.  ds_read_b128 costs the computing wave what global_load_dwordx4 does (ubench: +6 cycles per instruction).
# One iteration = PAIRL_BLOCKS blocks = one LDS buffer; s_barrier at the end of every iteration; the register ring
# (PAIRL_RING blocks) prefetches across the iteration boundary, which is why there are three buffers: the helper fills
# iteration k+2 while this wave works on k and prefetches from k+1.
PAIRL_BLOCKS = (16, 32)           # both are generated: k_scan_pairl_asm<BLOCKS>
PAIRL_RING = 8


class PairLPlan:
    R0 = 32
    O = R0 + 4 * PAIRL_RING
    PK = O + 16
    Z1 = PK + 8
    Z2 = Z1 + 1
    VCUR = Z2 + 1                 # lane * 16 + base of this iteration's LDS buffer
    VNXT = VCUR + 1               # ... of the next iteration's
    VOUT = VNXT + 1
    T0 = VOUT + 1
    LAST = T0


def pairl_iteration(first: bool, D: int):
    Q = PairLPlan
    RD = PAIRL_RING
    PAIRL_BUF = D * 1024
    out = []
    for i in range(D):
        g, half = i // 2, i % 2
        cur, prev = g & 1, (g & 1) ^ 1
        oc, op = Q.O + 8 * cur, Q.O + 8 * prev
        pk = Q.PK + 4 * prev
        slot = (i - 1) % RD                     # D % RD == 0: block i lives in ring slot i % RD
        j = i - 1 + RD                          # the block that slot receives
        src = (Q.VCUR, j) if j < D else (Q.VNXT, j - D)
        ld = f"ds_read_b128 v[{Q.R0 + 4 * slot}:{Q.R0 + 4 * slot + 3}], v{src[0]} offset:{src[1] * 1024}"
        nop = "s_nop 0"
        if first and i == 0:
            ld = nop                            # the prologue has filled the whole ring
        dead = first and g == 0
        gp = (g - 1) % (D // 2)
        if half == 0:
            fills = [ld,
                     nop if dead else f"v_cvt_pk_i16_i32 v{pk}, v{op}, v{op + 2}",
                     nop if dead else f"v_cvt_pk_i16_i32 v{pk + 1}, v{op + 1}, v{op + 3}",
                     nop if dead else f"v_cvt_pk_i16_i32 v{pk + 2}, v{op + 4}, v{op + 6}"]
        else:
            fills = [ld,
                     nop if dead else f"v_cvt_pk_i16_i32 v{pk + 3}, v{op + 5}, v{op + 7}",
                     nop if dead else f"global_store_dwordx4 v{Q.VOUT}, v[{pk}:{pk + 3}], %[pout] offset:{((gp % 8) - 4) * 1024}",
                     # ring slots (i+1) % RD, (i+2) % RD are next: their reads were issued RD-2 and RD-3 blocks ago; the reads
                     # issued since then are those of blocks i+4-RD .. i -> RD-3, minus one of margin
                     f"s_waitcnt lgkmcnt({RD - 4})"]
        for s_ in range(4):
            jj = 4 * half + s_
            xprev = oc + jj - 1 if jj > 0 else op + 7
            par = "e" if jj % 2 == 0 else "o"
            out.append(fills[s_])
            if fills[s_].startswith("global_store") and gp % 8 == 7:
                out.append(f"v_add_u32 v{Q.VOUT}, 0x2000, v{Q.VOUT}")
            k = Q.R0 + 4 * (i % RD) + PAIR_SLOT[s_]
            out.append(f"v_mad_i32_i24 v{Q.Z1}, %[co{par}], v{xprev}, v{k}")
            out.append(f"v_mul_i32_i24_dpp v{Q.Z2}, v{xprev}, %[cp{par}] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
            out.append(f"v_add_u32_sdwa v{oc + jj}, sext(v{Q.Z1}), sext(v{Q.Z2}) dst_sel:DWORD dst_unused:UNUSED_PAD "
                       f"src0_sel:WORD_1 src1_sel:WORD_1")
    # end of the iteration: the next buffer becomes the current one, the one after it (mod 3 buffers) the next; then meet the
    # helper, which has by now filled the buffer after that
    out += [f"v_mov_b32 v{Q.VCUR}, v{Q.VNXT}",
            f"v_add_u32 v{Q.T0}, {PAIRL_BUF}, v{Q.VNXT}",
            f"v_subrev_u32 v{Q.VNXT}, {3 * PAIRL_BUF}, v{Q.T0}",
            f"v_min_u32 v{Q.VNXT}, v{Q.VNXT}, v{Q.T0}",
            "s_barrier"]
    return out


def pairl_body(D: int):
    Q = PairLPlan
    RD = PAIRL_RING
    PAIRL_BUF = D * 1024
    assert D % RD == 0 and (D // 2) % 8 == 0 and D * 1024 <= 65536 - 1024 * RD
    b = [f"v_mov_b32 v{Q.O + 15}, %[x0]",
         f"v_mov_b32 v{Q.VCUR}, %[vlds]",
         f"v_add_u32 v{Q.VNXT}, {PAIRL_BUF}, v{Q.VCUR}",
         f"v_mov_b32 v{Q.VOUT}, %[vout]",
         "s_setprio 3",
         "s_barrier"]                             # the helper has filled iterations 0 and 1
    for n in range(RD):
        b.append(f"ds_read_b128 v[{Q.R0 + 4 * n}:{Q.R0 + 4 * n + 3}], v{Q.VCUR} offset:{n * 1024}")
    b.append("s_waitcnt lgkmcnt(0)")
    b += pairl_iteration(True, D)
    b += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_eq_u32 %[cnt], 0", "s_cbranch_scc1 2f", "1:"]
    b += pairl_iteration(False, D)
    b += ["s_sub_u32 %[cnt], %[cnt], 1", "s_cmp_lg_u32 %[cnt], 0", "s_cbranch_scc1 1b", "2:"]
    last_g = D // 2 - 1
    o, pk = Q.O + 8 * (last_g & 1), Q.PK + 4 * (last_g & 1)
    b += [f"v_cvt_pk_i16_i32 v{pk}, v{o}, v{o + 2}", f"v_cvt_pk_i16_i32 v{pk + 1}, v{o + 1}, v{o + 3}",
          f"v_cvt_pk_i16_i32 v{pk + 2}, v{o + 4}, v{o + 6}", f"v_cvt_pk_i16_i32 v{pk + 3}, v{o + 5}, v{o + 7}", "s_nop 1",
          f"global_store_dwordx4 v{Q.VOUT}, v[{pk}:{pk + 3}], %[pout] offset:{((last_g % 8) - 4) * 1024}",
          "s_waitcnt vmcnt(0) lgkmcnt(0)"]
    return b


def emit_pairl(D, name):
    b = pairl_body(D)
    clobbers = ", ".join(f'"v{r}"' for r in range(PairLPlan.R0, PairLPlan.LAST + 1))
    text = "\n".join(f'    "{l}\\n\\t"' for l in b)
    return (f"#define {name}_BODY \\\n{text.replace(chr(10), ' ' + chr(92) + chr(10))}\n"
            f'#define {name}_CLOBBERS {clobbers}, "memory", "scc"\n'), len(b)


def emit(name, P):
    b = body(P)
    clobbers = ", ".join(f'"v{r}"' for r in range(P.R0, P.LAST + 1))
    text = "\n".join(f'    "{l}\\n\\t"' for l in b)
    return (f"#define {name}_BODY \\\n{text.replace(chr(10), ' ' + chr(92) + chr(10))}\n"
            f'#define {name}_CLOBBERS {clobbers}, "memory", "scc"\n'), len(b)


def main():
    t32, n32 = emit("S5_SCAN_ASM", Plan(False))
    t16, n16 = emit("S5_SCAN16_ASM", Plan(True))
    t32w, n32w = emit("S5_SCAN32W_ASM", Plan(False, wide=True))
    tp, npair = emit_pair()
    tpl, npairl = "", []
    for D in PAIRL_BLOCKS:
        t, n = emit_pairl(D, f"S5_SCANPL{D}_ASM")
        tpl += t
        npairl.append(n)
    out = f"""// GENERATED by tools/gen_scan_asm.py -- do not edit.  DEPTH = {DEPTH}.
// Operands: [ca] [cb] [ka] [kb] [voff] [x0] VGPR inputs; [rin] [rout] 128-bit SGPR buffer descriptors;
// [stride] SGPR bytes per time block; [sld] [sst] [cnt] SGPR read-write (load / store offsets, iterations).
// S5_SCAN32W_ASM_BODY (exact 32-bit chain): [ca] [cb] are the plain multipliers, plus [sa] [sb] shifts, [ma] [mb] negation
// masks (0 / -1) and [oa] [ob] = mask & 1, all VGPR inputs; [ka] [kb] unused.
// S5_SCANP_ASM_BODY (pair kernel, {PAIR_DEPTH} blocks in flight): [coe] [cpe] [coo] [cpo] own / partner multipliers of even / odd steps,
// [vin] [vout] per-lane byte offsets (lane * 16 + 4096), [x0] VGPR inputs; [pin] [pout] 64-bit SGPR base addresses of this
// wave's run of the K stream / the packed state stream; [cnt] SGPR read-write (iterations of {PAIR_DEPTH} blocks).
#define S5_SCAN_ASM_DEPTH {DEPTH}
// S5_SCANPL_ASM_BODY (pair kernel fed from LDS): [vlds] = lane * 16 (byte address in LDS buffer 0) instead of [vin] / [pin];
// S5_SCANPL<D>_ASM_BODY: D blocks per iteration and LDS buffer, three buffers of D KB, one s_barrier per iteration (+ one up front).
#define S5_SCANP_ASM_DEPTH {PAIR_DEPTH}
{t32}{t16}{t32w}{tp}{tpl}"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparsernns_amd", "csrc", "scan_quad_asm.inc")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", os.path.normpath(path), n32, "+", n16, "+", n32w, "+", npair, "+", npairl, "instructions")


if __name__ == "__main__":
    main()
