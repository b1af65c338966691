#!/usr/bin/env python3
"""Turns the files `python sparseRNNs/fxprun.py --export ...` writes (reference: fxprun.py:475-495) into the
interchange pair `python -m sparsernns_amd.fxprun --model M.npz --meta M.json [--check-golden]` reads.

RUN THIS WHERE THE REFERENCE RUNS (it unpickles files that hold jax arrays, so it needs jax and must only be pointed at
files you produced yourself).  It is not exercised by this repository's tests: neither jax nor a trained checkpoint is
available offline.  With --check-golden the MI355X path is then compared bit for bit with the integer input / output
the reference itself recorded (fxpmodel_io.pkl) -- the way to pin parity against the real reference.

  python tools/convert_reference_export.py DATA_FOLDER OUT_PREFIX
"""
import json
import os
import pickle
import sys

import numpy as np


def flatten(tree, prefix=""):
    out = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            out.update(flatten(v, f"{prefix}{k}/"))
        else:
            out[f"{prefix}{k}"] = np.asarray(v)
    return out


def jsonable(tree):
    if isinstance(tree, dict):
        return {k: jsonable(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return [jsonable(v) for v in tree]
    if hasattr(tree, "item") and np.ndim(tree) == 0:
        return tree.item()
    return tree


def main():
    folder, out = sys.argv[1], sys.argv[2]
    with open(os.path.join(folder, "fxpmodel.pkl"), "rb") as f:
        model = pickle.load(f)  # {"params": ..., "qconfig": ...}  (FxpRegressionModel.export(), fxpmodel.py:1441-1458)
    arrays = {f"params/{k}": v.astype(np.int32) for k, v in flatten(model["params"]).items()}
    meta = {"export_qconfig": jsonable(model["qconfig"])}
    io_path = os.path.join(folder, "fxpmodel_io.pkl")
    if os.path.exists(io_path):
        with open(io_path, "rb") as f:
            io = pickle.load(f)  # FxpArrays: encoder input and decoder output of the exported run
        arrays["x"] = np.asarray(io["input"].data).astype(np.int32)
        arrays["y"] = np.asarray(io["output"].data).astype(np.int32)
        meta.update(x_bits=int(io["input"].bits), x_exp=int(io["input"].exp), y_bits=int(io["output"].bits),
                    y_exp=int(io["output"].exp))
    np.savez_compressed(out + ".npz", **arrays)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", out + ".npz", out + ".json")


if __name__ == "__main__":
    main()
