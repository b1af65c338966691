#!/usr/bin/env python3
"""sparsity_census.py -- is there anything to skip?  (CPU only: the oracle's integer traces of the synthetic N-DNS model)

BASELINE configs[2] (90 % unstructured weight sparsity, "CSR-gathered" projections) and configs[3] (ReLU activation-sparsity
gating) both pay only if whole units of work disappear.  On the fused path a unit of projection work is one MFMA operand
fragment (v_mfma_i32_32x32x32_i8: 32 output channels x 32 k of weights; 32 frames x 32 k of activations) or, for the
recurrence, one (sequence, state) chain step.  This tool counts how many such units are entirely zero

  * in the pruned weights (magnitude mask at 90 % per matrix, synth.make_model(sparsity=0.9)), and
  * in the activations the reference zeroes: the states after the complex ReLU (fxpmodel.py:740-742), the SSM output after
    the ReLU (:1125), the layer output after the ReLU (:1158-1159),

at the granularities a kernel could exploit: a weight fragment / row / column; an activation fragment (32 frames x 32 k), a
frame's whole operand row (what would let a frame skip the C projection), a 4-step block of one state.
Usage: python tools/sparsity_census.py [--dim-scale 0.5] [--frames 4x1024]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frag_zero(a: np.ndarray, r: int, c: int) -> float:
    """share of r x c fragments of the 2-D array a (padded with zeros to multiples) that are entirely zero"""
    R, C = (a.shape[0] + r - 1) // r * r, (a.shape[1] + c - 1) // c * c
    p = np.zeros((R, C), dtype=bool)
    p[:a.shape[0], :a.shape[1]] = a != 0
    return float(1.0 - p.reshape(R // r, r, C // c, c).any(axis=(1, 3)).mean())


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim-scale", type=float, default=0.5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seq-len", type=int, default=1024)
    args = ap.parse_args()
    from oracle import cref
    from oracle import fxp_oracle as O
    from sparsernns_amd import synth

    for sparsity in (0.0, 0.9):
        md, qc, dims = synth.make_model(args.dim_scale, sparsity=sparsity, calib_L=1024, state_headroom_bits=2 if sparsity else 1)
        om = O.RegressionModel(md, qc, dims["n_layers"])
        ex = om.export()
        print(f"=== dim_scale {args.dim_scale}, weight sparsity {sparsity:.0%}: H={dims['H']} P={dims['P']}")
        print("--- weights: share of zeros, and of all-zero MFMA fragments (32 channels x 32 k), 16x64 fragments, whole k-rows, whole channels")
        for i in range(dims["n_layers"]):
            mx = ex["params"]["encoder"][f"layers_{i}"]["mixer"]
            mats = {"B_re": np.asarray(mx["B_real"]), "B_im": np.asarray(mx["B_imag"]), "C_re": np.asarray(mx["C_real"]),
                    "C_im": np.asarray(mx["C_imag"]), "out2": np.asarray(ex["params"]["encoder"][f"layers_{i}"]["out2"]["weight"]).T}
            for name, wgt in mats.items():  # rows = output channels, columns = k
                print(f"layer {i} {name:5s} {wgt.shape!s:10s} zeros {np.mean(wgt == 0):6.1%}  32x32 {frag_zero(wgt, 32, 32):6.1%}  "
                      f"16x64 {frag_zero(wgt, 16, 64):6.1%}  k-columns {np.mean((wgt == 0).all(axis=0)):6.1%}  channels {np.mean((wgt == 0).all(axis=1)):6.1%}")
        enc = np.asarray(ex["params"]["encoder"]["encoder"]["weight"]).T
        dec = np.asarray(ex["params"]["decoder"]["weight"]).T
        for name, wgt in (("encoder", enc), ("decoder", dec)):
            print(f"{name:13s} {wgt.shape!s:10s} zeros {np.mean(wgt == 0):6.1%}  32x32 {frag_zero(wgt, 32, 32):6.1%}  16x64 {frag_zero(wgt, 16, 64):6.1%}")
        x = synth.make_input(args.batch, args.seq_len, dims["d_in"], seed=5)
        fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
        _, _, _, tr = cref.CModel(ex).forward(fx.data, fx.bits, fx.exp, trace=True)
        print(f"--- activations ({args.batch} x {args.seq_len} frames): non-zero share, all-zero frames (operand rows), all-zero 32-frame x 32-k MFMA fragments,")
        print("    all-zero 4-step blocks of one state (the recurrence kernels' item), frames with <= 8 non-zero entries")
        for i, t in enumerate(tr):
            re, im = t["xs_re"], t["xs_im"]
            keep = (re > 0) | ((re == 0) & (im > 0))           # complex ReLU, fxpmodel.py:30-45
            s = np.concatenate([np.where(keep, re, 0), np.where(keep, im, 0)], axis=-1).reshape(-1, 2 * dims["P"])
            y1 = np.maximum(t["ys"], 0).reshape(-1, dims["H"])
            out = np.maximum(t["residadd"], 0).reshape(-1, dims["H"])
            kb = keep.reshape(args.batch, args.seq_len // 4, 4, dims["P"]).any(axis=2)
            for name, a in (("xs_relu [re|im]", s), ("x1 = relu(ys)", y1), ("layer output", out)):
                nzrow = (a != 0).sum(axis=1)
                print(f"layer {i} {name:16s} non-zero {np.mean(a != 0):6.1%}  zero frames {np.mean(nzrow == 0):8.4%}  zero 32x32 fragments "
                      f"{frag_zero(a, 32, 32):8.4%}  frames with <= 8 non-zeros {np.mean(nzrow <= 8):7.3%}")
            print(f"layer {i} {'state 4-step blocks':16s} all four steps zeroed by the ReLU: {1.0 - kb.mean():6.1%} (the states themselves are never zero: the "
                  f"recurrence runs on the RAW state, :147-172)")


if __name__ == "__main__":
    main()
