#!/usr/bin/env python3
"""Generates tools/ubench_chain_gen.inc: instruction-issue experiments for the recurrence chain (diagnostic).

Every variant is a 16-step loop body for ONE wave per workgroup (the recurrence kernel's regime: a lone wave on its
SIMD), timed with s_memtime.  Registers are fixed: v10 = state x, v11..v14 temporaries, v2..v9 constants,
v[20:23] / v[24:27] load targets, v[28:29] packed outputs; s[8:11] / s[12:15] buffer descriptors, s16/s17 offsets.

Questions:
  * what does an independent / dependent VALU instruction cost a lone wave, and do fillers overlap the latency?
  * candidate step shapes:  CUR  mad, sdwa-add, 2 fillers, dpp-add          (round 1 kernel)
                            S1   mad, F, mul_dpp, sdwa-add(b), sdwa-add     (two lanes per state, b added separately)
                            S2   F, mad(K), mul_dpp, sdwa-add(hi, hi)       (two lanes per state, Bu folded into the addend)
"""
import os

SD = "dst_sel:DWORD dst_unused:UNUSED_PAD"
MAD = "v_mad_i32_i24 v11, v2, v10, v4"
MADK = "v_mad_i32_i24 v11, v2, v10, v20"           # addend from the ring register
SDWA_Q = f"v_add_u32_sdwa v13, sext(v11), v5 {SD} src0_sel:WORD_1 src1_sel:DWORD"
SDWA_Q_SAME = f"v_add_u32_sdwa v11, sext(v11), v5 {SD} src0_sel:WORD_1 src1_sel:DWORD"
MULDPP = "v_mul_i32_i24_dpp v12, v10, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
SDWA_X = f"v_add_u32_sdwa v10, sext(v12), v13 {SD} src0_sel:WORD_1 src1_sel:DWORD"
ADD_HH = f"v_add_u32_sdwa v10, sext(v11), sext(v12) {SD} src0_sel:WORD_1 src1_sel:WORD_1"
DPPADD = "v_add_u32_dpp v10, v11, v11 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
NOP = "s_nop 0"
SADD = "s_add_u32 s16, s16, s18"


def LOAD(i=0):
    return f"buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen offset:{1024 * i}"


def STORE(i=0):
    return f"buffer_store_dwordx2 v[28:29], v8, s[12:15], s17 offen offset:{512 * i}"


def STORE4(i=0):
    return f"buffer_store_dwordx4 v[24:27], v9, s[12:15], s17 offen offset:{1024 * i}"


PK = "v_cvt_pk_i16_i32 v28, v14, v15"
PK2 = "v_cvt_pk_i16_i32 v29, v14, v15"
INDV = "v_add_u32 v14, v2, v3"
WAIT = "s_waitcnt vmcnt(6)"


def rep(step, n=16):
    return [step] * n


VARIANTS = {}


def var(name, steps):
    """steps: list of 16 per-step instruction lists"""
    assert len(steps) == 16, name
    VARIANTS[name] = steps


# ---- issue model
var("dep1_add", rep(["v_add_u32 v10, v10, v2"]))
var("ind4_add", rep(["v_add_u32 v11, v10, v2", "v_add_u32 v12, v10, v3", "v_add_u32 v13, v10, v4", "v_add_u32 v14, v10, v5"]))
var("dep_plus_ind", rep(["v_add_u32 v10, v10, v2", INDV]))
var("dep_plus_nop", rep(["v_add_u32 v10, v10, v2", NOP]))
var("dep_plus_sadd", rep(["v_add_u32 v10, v10, v2", SADD]))
var("chains2", rep(["v_add_u32 v10, v10, v2", "v_add_u32 v11, v11, v2"]))
var("chains3", rep(["v_add_u32 v10, v10, v2", "v_add_u32 v11, v11, v2", "v_add_u32 v12, v12, v2"]))
var("dep1_mad", rep(["v_mad_i32_i24 v10, v2, v10, v4"]))
var("dep1_sdwa", rep([f"v_add_u32_sdwa v10, sext(v10), v5 {SD} src0_sel:WORD_1 src1_sel:DWORD"]))
var("dep1_muldpp_t", rep(["v_mul_i32_i24_dpp v10, v10, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"]))
var("dep2_mad_sdwa", rep([MAD, f"v_add_u32_sdwa v10, sext(v11), v5 {SD} src0_sel:WORD_1 src1_sel:DWORD"]))
# ---- round-1 shape
var("CUR_nop1", rep([MAD, SDWA_Q_SAME, "s_nop 1", DPPADD]))
var("CUR_2sadd", rep([MAD, SDWA_Q_SAME, SADD, SADD, DPPADD]))
# ---- S1: b added by its own sdwa-add
var("S1_F_first", rep([NOP, MAD, MULDPP, SDWA_Q, SDWA_X]))
var("S1_F_mid", rep([MAD, NOP, MULDPP, SDWA_Q, SDWA_X]))
var("S1_noF", rep([MAD, SDWA_Q, MULDPP, SDWA_X]))
# ---- S2: Bu folded into the addend
var("S2_F_first_nop", rep([NOP, MAD, MULDPP, ADD_HH]))
var("S2_F_mid_nop", rep([MAD, NOP, MULDPP, ADD_HH]))
var("S2_F_first_sadd", rep([SADD, MAD, MULDPP, ADD_HH]))
var("S2_F_first_valu", rep([INDV, MAD, MULDPP, ADD_HH]))
var("S2_noF_t", rep([MAD, MULDPP, ADD_HH]))
var("S2_2F", rep([NOP, NOP, MAD, MULDPP, ADD_HH]))
# S2 with the real filler mix: per 4-step block  load, pack, pack, store; per 4 blocks one offset add each + one wait
def blk(i):
    return [[LOAD(i), MADK, MULDPP, ADD_HH], [PK, MADK, MULDPP, ADD_HH], [PK2, MADK, MULDPP, ADD_HH], [STORE(i), MADK, MULDPP, ADD_HH]]


var("S2_mem_exact4", sum((blk(i) for i in range(4)), []))
blk_x = [list(s) for s in sum((blk(i) for i in range(4)), [])]
blk_x[3].insert(1, SADD)
blk_x[7].insert(1, "s_add_u32 s17, s17, s19")
blk_x[11].insert(1, WAIT)
var("S2_mem_plus3", blk_x)
# the same without packing (int32 states out): load, store + 2 spare slots per block
def blk32(i):
    return [[LOAD(i), MADK, MULDPP, ADD_HH], [NOP, MADK, MULDPP, ADD_HH], [SADD if i == 3 else NOP, MADK, MULDPP, ADD_HH],
            [STORE4(i), MADK, MULDPP, ADD_HH]]


var("S2_mem_i32out", sum((blk32(i) for i in range(4)), []))
# S1 with the real filler mix (one filler slot per step as well)
def blk1(i):
    return [[LOAD(i), MAD, MULDPP, SDWA_Q, SDWA_X], [PK, MAD, MULDPP, SDWA_Q, SDWA_X], [PK2, MAD, MULDPP, SDWA_Q, SDWA_X],
            [STORE(i), MAD, MULDPP, SDWA_Q, SDWA_X]]


var("S1_mem_exact4", sum((blk1(i) for i in range(4)), []))


# ---- round 2: what does ONE memory instruction per 4-step block cost, by kind?  Base = S2 with nop fillers (18.0).
def pattern(fills, per_step=(MAD, MULDPP, ADD_HH)):
    """16 steps; fills = list of 16 filler instructions (one per step, first slot)"""
    return [[f] + list(per_step) for f in fills]


def every4(x, others=NOP):
    return [x if i % 4 == 0 else others for i in range(16)]


LDS_RD = "ds_read_b128 v[20:23], v7"
LDS_RD64 = "ds_read_b64 v[20:21], v7"
LDS_WR64 = "ds_write_b64 v7, v[28:29]"
LDS_WR128 = "ds_write_b128 v7, v[24:27]"
var("F_load_x4", pattern(every4("buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen")))
var("F_load_x2", pattern(every4("buffer_load_dwordx2 v[20:21], v8, s[8:11], s16 offen")))
var("F_load_x1", pattern(every4("buffer_load_dword v20, v8, s[8:11], s16 offen")))
var("F_load_x4_off", pattern(every4("buffer_load_dwordx4 v[20:23], off, s[8:11], s16")))
var("F_gload_x4", pattern(every4("global_load_dwordx4 v[20:23], v9, s[8:9]")))
var("F_store_x4", pattern(every4("buffer_store_dwordx4 v[24:27], v9, s[12:15], s17 offen")))
var("F_store_x2", pattern(every4("buffer_store_dwordx2 v[28:29], v8, s[12:15], s17 offen")))
var("F_store_x1", pattern(every4("buffer_store_dword v28, v8, s[12:15], s17 offen")))
var("F_store_x2_off", pattern(every4("buffer_store_dwordx2 v[28:29], off, s[12:15], s17")))
var("F_gstore_x2", pattern(every4("global_store_dwordx2 v8, v[28:29], s[12:13]")))
var("F_cvt", pattern(every4(PK)))
var("F_ds_rd128", pattern(every4(LDS_RD)))
var("F_ds_rd64", pattern(every4(LDS_RD64)))
var("F_ds_wr64", pattern(every4(LDS_WR64)))
var("F_ds_wr128", pattern(every4(LDS_WR128)))
var("F_ds_rd128_wr64", pattern([LDS_RD, NOP, NOP, LDS_WR64] * 4))
var("F_ds_rd128_pk_pk_wr64", pattern([LDS_RD, PK, PK2, LDS_WR64] * 4))
var("F_ld_st_x4x2", pattern(["buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen", NOP, NOP, "buffer_store_dwordx2 v[28:29], v8, s[12:15], s17 offen"] * 4))
var("F_ld_only_all", pattern(["buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen"] * 16))
var("F_barrier_per16", pattern(["s_barrier"] + [NOP] * 15))
var("F_waitcnt_each4", pattern(every4("s_waitcnt vmcnt(0) lgkmcnt(0)")))
# a load every 8 steps, a store every 8 steps (wider items)
var("F_ld4_st8", pattern((["buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen", NOP, NOP, NOP,
                           "buffer_load_dwordx4 v[20:23], v9, s[8:11], s16 offen", NOP, NOP, "buffer_store_dwordx4 v[24:27], v9, s[12:15], s17 offen"]) * 2))
# round-1 shape with its real fillers, for reference
var("CUR_ld_st", [[MAD, SDWA_Q_SAME, ("buffer_load_dwordx2 v[20:21], v8, s[8:11], s16 offen" if i % 4 == 0 else SADD if i % 4 == 1 else "buffer_store_dwordx2 v[28:29], v8, s[12:15], s17 offen" if i % 4 == 2 else NOP), (PK if i % 4 < 2 else NOP), DPPADD] for i in range(16)])


def main():
    out = ["// GENERATED by tools/gen_ubench_chain.py -- do not edit."]
    names = []
    for name, steps in VARIANTS.items():
        lines = [ins for st in steps for ins in st]
        body = " \\\n".join(f'    "{l}\\n\\t"' for l in lines)
        out.append(f"#define UB_BODY_{name} \\\n{body}\n")
        names.append((name, len(lines)))
    out.append("#define UB_ALL(X) " + " ".join(f"X({n}, {k})" for n, k in names))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_chain_gen.inc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", path, len(names), "variants")


if __name__ == "__main__":
    main()
