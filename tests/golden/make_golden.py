#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (committed; rerun only when the spec changes).

PARITY UNPINNED: the reference cannot run here (no JAX), so these vectors come from
the CPU oracle (oracle/fxp_oracle.py), which both oracle halves and the HIP path must then reproduce
bit for bit.  They pin the semantics against regressions; they are not reference outputs.

Each fixture holds: the float modeldict leaves + fxp_qconfig (json), the integer input, every
intermediate the oracle names, and the output.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from oracle import fxp_oracle as O  # noqa: E402
from sparsernns_amd import synth  # noqa: E402

CASES = {
    "tiny_a": dict(make=dict(dims=synth.tiny_dims(), seed=7), B=2, L=24, scale=1.0),
    "tiny_b_bnscale": dict(make=dict(dims=synth.tiny_dims(H=12, P=6, d_in=7, d_out=9, n_layers=2), bn_scale_bias=True,
                                     input_scale=30.0, seed=11), B=2, L=20, scale=30.0),
    "ndns05_short": dict(make=dict(dim_scale=0.5, seed=1919), B=1, L=16, scale=1.0),
}


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(flatten(v, f"{prefix}{k}/"))
        else:
            out[f"{prefix}{k}"] = v
    return out


def main():
    for name, c in CASES.items():
        md, qc, dims = synth.make_model(**c["make"])
        x = synth.make_input(c["B"], c["L"], dims["d_in"], seed=3, scale=c["scale"])
        fx = O.from_fp(x, qc["encoder"]["inp_bits"], qc["encoder"]["inp_exp"], True, O.FLOOR)
        model = O.RegressionModel(md, qc, dims["n_layers"])
        inter = {}
        y = model(fx, inter)
        small = dims["H"] <= 16
        # float parameters only for the small models (they pin the float -> int setup path as well)
        arrays = {f"md/{k}": np.asarray(v) for k, v in flatten(md).items()} if small else {}
        ex = model.export()
        for k, v in flatten(ex["params"]).items():  # the integer model, in the smallest dtype that holds it
            v = np.asarray(v)
            dt = np.int8 if np.abs(v).max(initial=0) < 128 else (np.int16 if np.abs(v).max(initial=0) < 32768 else np.int32)
            arrays[f"params/{k}"] = v.astype(dt)
        arrays["x"] = fx.data.astype(np.int16)
        arrays["y"] = y.data.astype(np.int16)
        meta = dict(qconfig=qc, export_qconfig=ex["qconfig"], dims=dims, x_bits=fx.bits, x_exp=fx.exp, y_bits=y.bits,
                    y_exp=y.exp, inter={})
        keep = None if small else ("mixer.ys", "residadd", "mixer.xs_re", "mixer.Bu_im")
        for k, v in O.flatten_intermediates(inter).items():
            if keep is not None and not k.endswith(keep):
                continue
            arrays[f"inter/{k}"] = v.data
            meta["inter"][k] = [v.bits, v.exp]
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **arrays)
        with open(os.path.join(HERE, f"{name}.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(name, "->", sum(a.nbytes for a in arrays.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
