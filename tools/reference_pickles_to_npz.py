#!/usr/bin/env python3
"""Turns the reference's pickles into the interchange files this repository reads -- RUN IT WHERE THE REFERENCE RUNS.

  calib  DATA_FOLDER OUT_PREFIX   sc_calibrated_params.pkl + sc_cal_stats.pkl (convert.py:963,971)
                                  -> OUT_PREFIX.params.npz + OUT_PREFIX.stats.npz, the inputs of
                                     `python -m sparsernns_amd.fxprun --params ... --stats ...` (sparsernns_amd/fxputils.py)
  export DATA_FOLDER OUT_PREFIX   fxpmodel.pkl [+ fxpmodel_io.pkl] written by `fxprun.py --export` (fxprun.py:475-495)
                                  -> OUT_PREFIX.npz + OUT_PREFIX.json, the inputs of
                                     `python -m sparsernns_amd.fxprun --model ... --meta ... [--check-golden]`: with the
                                     recorded integer input / output this pins the MI355X path against the real reference

The pickles hold trees of JAX / NumPy arrays (and, for `export`, FxpArray dataclasses).  They are read with an unpickler
that only reconstructs those: dicts, lists, tuples, numbers, strings, NumPy arrays / scalars / dtypes, JAX arrays, and the
reference's FxpArray.  Any other global in the stream (which is how a pickle runs code) raises.  Only point it at files you
produced yourself all the same.  Needs whatever wrote the arrays (jax, where the arrays are jax arrays) to be importable.
"""
import io
import json
import os
import pickle
import sys

import numpy as np

ALLOWED_PREFIXES = ("numpy.", "numpy", "jax._src.array", "jax._src.core", "jax.numpy", "jaxlib.", "ml_dtypes.")
ALLOWED_EXACT = {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str", "bytes",
                                           "bytearray", "slice", "range", "object")} | {
    ("collections", "OrderedDict"), ("copyreg", "_reconstructor"), ("sparseRNNs.fxparray", "FxpArray"),
    ("sparseRNNs.fxparray", "ComplexFxpArray"), ("sparseRNNs.fxparray", "RoundingMode")}


class ArraysOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in ALLOWED_EXACT or any(module == p.rstrip(".") or module.startswith(p) for p in ALLOWED_PREFIXES):
            if module.startswith("numpy") and name in ("load", "loads", "fromfile", "memmap", "DataSource", "save", "savez"):
                raise pickle.UnpicklingError(f"refusing {module}.{name}")
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle a reference to {module}.{name}: only array trees are read")


def load_arrays_only(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return ArraysOnlyUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as f:
        return ArraysOnlyUnpickler(f).load()


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


def convert_calib(folder, out):
    for name, suffix in (("sc_calibrated_params.pkl", ".params.npz"), ("sc_cal_stats.pkl", ".stats.npz")):
        tree = load_arrays_only(os.path.join(folder, name))
        np.savez_compressed(out + suffix, **flatten(tree))
        print("wrote", out + suffix)


def convert_export(folder, out):
    model = load_arrays_only(os.path.join(folder, "fxpmodel.pkl"))  # {"params", "qconfig"}: FxpRegressionModel.export(), fxpmodel.py:1441-1458
    arrays = {f"params/{k}": v.astype(np.int32) for k, v in flatten(model["params"]).items()}
    meta = {"export_qconfig": jsonable(model["qconfig"])}
    io_path = os.path.join(folder, "fxpmodel_io.pkl")
    if os.path.exists(io_path):
        rec = load_arrays_only(io_path)  # FxpArrays: encoder input and decoder output of the exported run
        arrays["x"] = np.asarray(rec["input"].data).astype(np.int32)
        arrays["y"] = np.asarray(rec["output"].data).astype(np.int32)
        meta.update(x_bits=int(rec["input"].bits), x_exp=int(rec["input"].exp), y_bits=int(rec["output"].bits),
                    y_exp=int(rec["output"].exp))
    np.savez_compressed(out + ".npz", **arrays)
    with open(out + ".json", "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", out + ".npz", out + ".json")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3 or argv[0] not in ("calib", "export"):
        print(__doc__)
        return 2
    (convert_calib if argv[0] == "calib" else convert_export)(argv[1], argv[2])
    return 0


if __name__ == "__main__":
    sys.exit(main())
