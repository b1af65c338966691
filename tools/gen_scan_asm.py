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
    out = f"""// GENERATED by tools/gen_scan_asm.py -- do not edit.  DEPTH = {DEPTH}.
// Operands: [ca] [cb] [ka] [kb] [voff] [x0] VGPR inputs; [rin] [rout] 128-bit SGPR buffer descriptors;
// [stride] SGPR bytes per time block; [sld] [sst] [cnt] SGPR read-write (load / store offsets, iterations).
// S5_SCAN32W_ASM_BODY (exact 32-bit chain): [ca] [cb] are the plain multipliers, plus [sa] [sb] shifts, [ma] [mb] negation
// masks (0 / -1) and [oa] [ob] = mask & 1, all VGPR inputs; [ka] [kb] unused.
#define S5_SCAN_ASM_DEPTH {DEPTH}
{t32}{t16}{t32w}"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparsernns_amd", "csrc", "scan_quad_asm.inc")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", os.path.normpath(path), n32, "+", n16, "+", n32w, "instructions")


if __name__ == "__main__":
    main()
