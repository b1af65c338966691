#!/usr/bin/env python3
"""Turns the reference's pickles into the interchange files this repository reads -- RUN IT WHERE THE REFERENCE RUNS.

  calib  DATA_FOLDER OUT_PREFIX   sc_calibrated_params.pkl + sc_cal_stats.pkl (convert.py:963,971)
                                  -> OUT_PREFIX.params.npz + OUT_PREFIX.stats.npz, the inputs of
                                     `python -m sparsernns_amd.fxprun --params ... --stats ...` (sparsernns_amd/fxputils.py)
  export DATA_FOLDER OUT_PREFIX   fxpmodel.pkl [+ fxpmodel_io.pkl] written by `fxprun.py --export` (fxprun.py:475-495)
                                  -> OUT_PREFIX.npz + OUT_PREFIX.json, the inputs of
                                     `python -m sparsernns_amd.fxprun --model ... --meta ... [--check-golden]`: with the
                                     recorded integer input / output this pins the MI355X path against the real reference

  acts   ACTIVATIONS.pkl OUT.npz  activations_fp.pkl (the float model's sown intermediates, convert.py) -> the npz tree
                                  `fxprun --verify --activations OUT.npz` compares the fixed-point stages with
                                  (sparsernns_amd/fxpreporter.py): first recorded call, batch item 0 of every stage

The pickles hold trees of JAX / NumPy arrays (and, for `export`, FxpArray dataclasses).  They are read with an unpickler
that only reconstructs those: dicts, lists, tuples, numbers, strings, NumPy arrays / scalars / dtypes, JAX arrays, and the
reference's FxpArray -- an exact (module, name) set.  Any other global in the stream (which is how a pickle runs code) raises.  Only point it at files you
produced yourself all the same.  Needs whatever wrote the arrays (jax, where the arrays are jax arrays) to be importable.
"""
import io
import json
import os
import pickle
import sys

import numpy as np

# Exactly the globals that rebuilding an array tree needs -- (module, name) pairs, no prefixes: a prefix such as "numpy."
# would also admit numpy.testing / numpy.distutils callables, i.e. ways to run code.
_NP = ("numpy.core", "numpy._core")  # NumPy 1.x / 2.x module paths of the same functions
ALLOWED = {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "complex", "bool", "str", "bytes",
                                     "bytearray", "slice", "range", "object")} | {
    ("collections", "OrderedDict"), ("copyreg", "_reconstructor"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    # what jax.Array.__reduce__ emits around the NumPy reconstruction of its value (jax/_src/array.py)
    ("jax._src.array", "_reconstruct_array"),
    # the reference's own containers (`export` mode; importing them needs the reference on the path)
    ("sparseRNNs.fxparray", "FxpArray"), ("sparseRNNs.fxparray", "ComplexFxpArray"), ("sparseRNNs.fxparray", "RoundingMode"),
} | {(f"{m}.multiarray", n) for m in _NP for n in ("_reconstruct", "scalar")} | {(f"{m}.numeric", "_frombuffer") for m in _NP}


class ArraysOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in ALLOWED:
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


def convert_acts(path, out):
    """activations_fp.pkl: {"encoder": {"layers_i": {"input": [arr (B,L,H)], "pre_s5": [...], "pre_C": [...], "pre_GLU": [...],
    "mixer": {"B_bar": [..], "__call__": [(ys, xs)]}, "out2": {"__call__": [...]}, "drop": {"__call__": [x1, post_glu]},
    "__call__": [...]}}, "__call__": [...]} as fxprun.py:583-727 indexes it -> one array per stage for sequence 0."""
    tree = load_arrays_only(path)

    def first(v):  # first recorded call
        return v[0] if isinstance(v, (list, tuple)) else v

    def seq0(a):  # batch item 0
        a = np.asarray(a)
        return a[0] if a.ndim >= 3 else a

    flat = {"__call__": seq0(first(tree["__call__"]))}
    for name, lay in tree["encoder"].items():
        if not name.startswith("layers_"):
            continue
        pre = f"encoder/{name}/"
        for k in ("input", "pre_s5", "pre_C", "pre_GLU", "__call__"):
            flat[pre + k] = seq0(first(lay[k]))
        flat[pre + "mixer/B_bar"] = np.asarray(first(lay["mixer"]["B_bar"]))
        if flat[pre + "mixer/B_bar"].ndim == 3:
            flat[pre + "mixer/B_bar"] = flat[pre + "mixer/B_bar"][0]
        flat[pre + "mixer/__call__"] = seq0(first(first(lay["mixer"]["__call__"])))  # [0][0]: ys of the (ys, xs) pair
        flat[pre + "out2/__call__"] = seq0(first(lay["out2"]["__call__"]))
        drop = lay["drop"]["__call__"]
        flat[pre + "post_GLU"] = seq0(drop[1]) if isinstance(drop, (list, tuple)) and len(drop) == 2 else flat[pre + "pre_GLU"]  # fxprun.py:688-698
    np.savez_compressed(out, **flat)
    print("wrote", out)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3 or argv[0] not in ("calib", "export", "acts"):
        print(__doc__)
        return 2
    {"calib": convert_calib, "export": convert_export, "acts": convert_acts}[argv[0]](argv[1], argv[2])
    return 0


if __name__ == "__main__":
    sys.exit(main())
