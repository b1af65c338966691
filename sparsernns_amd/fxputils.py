"""From a calibrated checkpoint to ``modeldict`` + ``fxp_qconfig`` (host side, NumPy; runs once per model).

Counterpart of the reference's ``sparseRNNs/fxputils.py``: ``load_modeldict`` (:121-134), ``create_fxp_qconfig``
(:351-401) and ``add_target_bits_exp`` (:453-786), i.e. what ``fxprun.py:294-397`` does between reading
``sc_calibrated_params.pkl`` / ``sc_cal_stats.pkl`` and building ``FxpRegressionModel``.  The reference keeps its
trees as pickles of JAX arrays; here the interchange is an ``.npz`` whose keys are the ``/``-joined tree paths
(``save_tree_npz`` / ``load_tree_npz``; INTEGRATION.md section 6 shows the five-line export a maintainer runs where
JAX exists).  Nothing here touches the GPU.

The rules, stated once (the reference spells them out per tensor):

* every leaf whose key contains ``scale`` becomes ``log2(scale)`` (:128-131) -- this includes BatchNorm's ``scale``
  parameter, whose NaNs (log2 of a negative) ``add_target_bits_exp`` later replaces by 1.0 (:711-731): a quirk of the
  reference that the fixed-point model then consumes as given;
* every ``*observer`` dict gains ``absmax = max |min, max|`` and ``intbits = ceil(log2 absmax)`` (:68-80);
* per tensor: ``intbits = max(0, ceil(log2 absmax)) + [absmax is a power of two]`` (:137-142),
  ``fracbits = -ceil(log2 scale)`` (:178,249), sign information from the observer range (:145-153);
* shared exponents (``agg="max"``): the maximum over layers of every field (:288-348); a field that differs in sign
  between layers gets a sign bit (:281-286);
* target widths: ``exp = min(fracbits, bits - 1 - intbits)`` (:404-450, :553-586); the gate result gets
  ``res_exp = bits - 1 - (l_intbits + r_intbits)`` (:530-537); BatchNorm tensors ``exp = bits - 1 - max(0, ceil(log2
  absmax))`` with ``invsq_var = rsqrt(var + 1e-5)`` (:636-752).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

F32 = np.float32
BN_EPS = 1e-5

W8A16 = dict(non_ssm_w=8, non_ssm_b=16, non_ssm_act=16, ssm_w=8, ssm_act=16)


def precisions_for(quantization: str) -> Dict[str, int]:
    """fxprun.py:302-308: weights 8 bit; biases and activations 16 bit when the recipe name contains "a16"."""
    act = 16 if "a16" in quantization else 8
    return dict(non_ssm_w=8, non_ssm_b=act, non_ssm_act=act, ssm_w=8, ssm_act=act)


# --------------------------------------------------------------------------------------
# trees <-> npz
# --------------------------------------------------------------------------------------
def flatten_tree(tree: dict, prefix: str = "") -> Dict[str, np.ndarray]:
    out = {}
    for k, v in tree.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_tree(v, key))
        else:
            out[key] = np.asarray(v)
    return out


def unflatten_tree(flat: Dict[str, np.ndarray]) -> dict:
    tree: dict = {}
    for key, v in flat.items():
        node = tree
        parts = key.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


def save_tree_npz(path: str, tree: dict) -> None:
    np.savez_compressed(path, **flatten_tree(tree))


def load_tree_npz(path: str) -> dict:
    with np.load(path, allow_pickle=False) as z:
        return unflatten_tree({k: z[k] for k in z.files})


# --------------------------------------------------------------------------------------
# load_modeldict
# --------------------------------------------------------------------------------------
def merge_params_and_stats(params, stats):
    """Union of the two trees; where both hold a leaf the parameter wins (fxputils.py:12-64)."""
    if isinstance(params, dict) and isinstance(stats, dict):
        return {k: (merge_params_and_stats(params[k], stats[k]) if k in params and k in stats
                    else params.get(k, stats.get(k))) for k in list(params) + [k for k in stats if k not in params]}
    return params if params is not None else stats


def _map_leaves(tree, fn, key=""):
    if isinstance(tree, dict):
        return {k: _map_leaves(v, fn, k) for k, v in tree.items()}
    return fn(key, tree)


def _observer_summary(tree, key=""):
    if isinstance(tree, dict):
        if key.endswith("observer"):
            first = next(iter(tree)).split("_")[0]
            stem = "" if first == "observer" else f"{first}_"
            lo, hi = np.asarray(tree[f"{stem}observer_min"]), np.asarray(tree[f"{stem}observer_max"])
            out = dict(tree)
            out["absmax"] = np.maximum(np.abs(lo).max(), np.abs(hi).max()).astype(F32)
            with np.errstate(divide="ignore"):
                out["intbits"] = np.ceil(np.log2(out["absmax"])).astype(int)
            return out
        return {k: _observer_summary(v, k) for k, v in tree.items()}
    return tree


def load_modeldict(params: dict, stats: dict) -> dict:
    """params / stats: the trees of ``sc_calibrated_params.pkl`` / ``sc_cal_stats.pkl`` (as dicts of arrays, e.g. from
    ``load_tree_npz``).  Returns the ``modeldict`` the fixed-point model consumes (fxputils.py:121-134)."""
    md = merge_params_and_stats(params, stats)
    with np.errstate(invalid="ignore", divide="ignore"):
        md = _map_leaves(md, lambda k, v: np.log2(np.asarray(v, dtype=F32)) if "scale" in k else v)
    return _observer_summary(md)


# --------------------------------------------------------------------------------------
# create_fxp_qconfig
# --------------------------------------------------------------------------------------
def get_intbits(absmax: float) -> int:
    l2 = math.log2(absmax)
    return max(0, math.ceil(l2)) + (1 if round(l2) == l2 else 0)


def get_sign(lo: float, hi: float):
    """1: needs a sign bit; 0.75 / 0.25: none, always positive / always negative (fxputils.py:145-153)."""
    lo, hi = float(lo), float(hi)
    if lo * hi < 0:
        return 1
    return 0.75 if (lo, hi) == (abs(lo), abs(hi)) else 0.25


def _at(tree: dict, path: str):
    for k in path.split("/"):
        tree = tree[k]
    return tree


def _quant_entry(node: dict) -> dict:
    """One observed tensor of the SSM / gate: {scale (log2), observer{observer_min, observer_max, absmax}}."""
    obs = node["observer"]
    absmax = float(obs["absmax"])
    return dict(absmax=absmax, intbits=get_intbits(absmax), signbits=get_sign(obs["observer_min"], obs["observer_max"]),
                fracbits=-int(math.ceil(float(node["scale"]))))


SSM_WEIGHTS = dict(A_re="quant_A/quant_real", A_im="quant_A/quant_imag", B_re="quant_B/quant_real", B_im="quant_B/quant_imag",
                   C_re="quant_C/quant_real", C_im="quant_C/quant_imag", D="quant_D")
SSM_ACTS = dict(u="quant_ut", Bu_re="quant_But/quant_real", Bu_im="quant_But/quant_imag", x_re="quant_xt/quant_real",
                x_im="quant_xt/quant_imag", y="quant_yt")


def ssm_qconfig(mixer: dict) -> dict:
    return dict(weights={k: _quant_entry(_at(mixer, p)) for k, p in SSM_WEIGHTS.items()},
                activations={k: _quant_entry(_at(mixer, p)) for k, p in SSM_ACTS.items()})


def multgate_qconfig(gate: dict) -> dict:
    out = {}
    for side, name in (("l", "quant_left"), ("r", "quant_right")):
        e = _quant_entry(gate[name])
        out.update({f"{side}_absmax": e["absmax"], f"{side}_signbits": e["signbits"], f"{side}_intbits": e["intbits"],
                    f"{side}_fracbits": e["fracbits"]})
    return out


def dense_qconfig(dense: dict) -> dict:
    w, b = np.asarray(dense["kernel"]), np.asarray(dense["bias"])
    fields = dict(
        b=(float(np.abs(b).max()), get_sign(b.min(), b.max()), dense["act_scale"]),  # the bias shares the input's scale (:266)
        w=(float(np.abs(w).max()), get_sign(w.min(), w.max()), dense["weight_scale"]),
        inp=(float(dense["input_observer"]["absmax"]), get_sign(dense["input_observer"]["input_observer_min"],
                                                                dense["input_observer"]["input_observer_max"]), dense["act_scale"]),
        out=(float(dense["output_observer"]["absmax"]), get_sign(dense["output_observer"]["output_observer_min"],
                                                                 dense["output_observer"]["output_observer_max"]), dense["out_scale"]))
    out = {}
    for name, (absmax, sign, log2_scale) in fields.items():
        out.update({f"{name}_absmax": absmax, f"{name}_signbit": sign, f"{name}_intbits": get_intbits(absmax),
                    f"{name}_fracbits": -int(math.ceil(float(log2_scale)))})
    return out


def _layer_keys(tree: dict) -> Iterable[str]:
    return sorted((k for k in tree if k.startswith("layers_")), key=lambda s: int(s.split("_")[1]))


def _join(per_layer: Dict[str, dict], agg: str, sign_suffix_from: int) -> dict:
    """{layer: {field: value}} -> {field: max over layers} (agg="max") or {field: sorted distinct values} ("set")."""
    first = next(iter(per_layer.values()))
    out = {}
    for f in first:
        vals = sorted(set(per_layer[l][f] for l in per_layer))
        if agg == "set":
            out[f] = vals
        elif len(vals) == 1:
            out[f] = vals[0]
        else:  # layers that disagree on the sign need a sign bit (:281-286)
            out[f] = 1 if f[sign_suffix_from:] == "signbits" else max(vals)
    return out


def create_fxp_qconfig(modeldict: dict, agg: Optional[str] = "max"):
    """agg "max" / "set": returns (per_layer, joined) like the reference (:351-401); agg None: the per-layer tree in the
    ``--separate_exponents`` layout ``blocks/layers_i/{ssm, multgate, out2}``."""
    if agg not in ("max", "set", None):
        raise ValueError("agg must be None, 'max' or 'set'")
    enc = modeldict["encoder"]
    layers = list(_layer_keys(enc))
    per = dict(encoder=dense_qconfig(enc["encoder"]),
               blocks=dict(ssm={l: ssm_qconfig(enc[l]["mixer"]) for l in layers},
                           multgate={l: multgate_qconfig(enc[l]["mult_gate"]) for l in layers},
                           out2={l: dense_qconfig(enc[l]["out2"]) for l in layers}),
               decoder=dense_qconfig(modeldict["decoder"]))
    if agg is None:
        return dict(encoder=per["encoder"], decoder=per["decoder"],
                    blocks={l: dict(ssm=per["blocks"]["ssm"][l], multgate=per["blocks"]["multgate"][l],
                                    out2=per["blocks"]["out2"][l]) for l in layers})
    ssm = per["blocks"]["ssm"]
    joined = dict(encoder=per["encoder"], decoder=per["decoder"], blocks=dict(
        out2=_join(per["blocks"]["out2"], agg, 2), multgate=_join(per["blocks"]["multgate"], agg, 2),
        ssm={grp: {t: _join({l: ssm[l][grp][t] for l in layers}, agg, 0) for t in ssm[layers[0]][grp]}
             for grp in ("weights", "activations")}))
    return per, joined


# --------------------------------------------------------------------------------------
# add_target_bits_exp
# --------------------------------------------------------------------------------------
def _fit(fracbits: int, intbits: int, bits: int) -> int:
    return min(fracbits, bits - 1 - intbits)


def _dense_targets(q: dict, prec: dict) -> dict:
    for name, bits in (("w", prec["non_ssm_w"]), ("b", prec["non_ssm_b"]), ("inp", prec["non_ssm_act"]), ("out", prec["non_ssm_act"])):
        q[f"{name}_bits"] = bits
        q[f"{name}_exp"] = _fit(q[f"{name}_fracbits"], q[f"{name}_intbits"], bits)
    return q


def _gate_targets(q: dict, prec: dict) -> dict:
    bits = prec["non_ssm_act"]
    for side in ("l", "r"):
        q[f"{side}_bits"] = bits
        q[f"{side}_exp"] = _fit(q[f"{side}_fracbits"], q[f"{side}_intbits"], bits)
    q["res_bits"] = bits
    q["res_exp"] = bits - 1 - (q["l_intbits"] + q["r_intbits"])  # no statistics are collected for the product (:527)
    return q


def _ssm_targets(q: dict, prec: dict) -> dict:
    for t, e in q["weights"].items():
        e["bits"] = prec["ssm_act"] if t in ("A_re", "A_im") else prec["ssm_w"]  # Lambda-bar gets the activation width (:557-562)
        e["exp"] = _fit(e["fracbits"], e["intbits"], e["bits"])
    for e in q["activations"].values():
        e["bits"] = prec["ssm_act"]
        e["exp"] = _fit(e["fracbits"], e["intbits"], e["bits"])
    return q


def _norm_targets(norms: Iterable[dict], prec: dict) -> dict:
    """norms: the ``norm`` sub-dicts (mean, var, [scale], [bias]) whose tensors share one exponent each."""
    norms = list(norms)
    bits = prec["non_ssm_act"]

    def entry(arrays):
        ib = max(0, int(math.ceil(math.log2(max(float(np.abs(np.asarray(a)).max()) for a in arrays)))))
        return dict(intbits=ib, exp=bits - 1 - ib, bits=bits)

    out = dict(mean=entry(n["mean"] for n in norms), var=entry(n["var"] for n in norms),
               invsq_var=entry((F32(1.0) / np.sqrt(np.asarray(n["var"], dtype=F32) + F32(BN_EPS))) for n in norms))
    for k in ("scale", "bias"):
        if k in norms[0]:
            for n in norms:  # log2 of a non-positive scale is NaN: the reference substitutes 1.0, in place (:711-731)
                n[k] = np.where(np.isnan(np.asarray(n[k], dtype=F32)), F32(1.0), np.asarray(n[k], dtype=F32))
            out[k] = entry(n[k] for n in norms)
    return out


def add_target_bits_exp(modeldict: dict, fxp_qconfig: dict, precisions: Dict[str, int], shared_exp: bool = True) -> dict:
    """Adds ``*_bits`` / ``*_exp`` to every tensor of ``fxp_qconfig`` (in place, returned) and the ``norm`` block.
    ``modeldict``'s BatchNorm scale / bias NaNs are replaced by 1.0 as a side effect, as in the reference."""
    enc = modeldict["encoder"]
    layers = list(_layer_keys(enc))
    _dense_targets(fxp_qconfig["encoder"], precisions)
    _dense_targets(fxp_qconfig["decoder"], precisions)
    blocks = [fxp_qconfig["blocks"]] if shared_exp else [fxp_qconfig["blocks"][l] for l in layers]
    for b in blocks:
        _dense_targets(b["out2"], precisions)
        _gate_targets(b["multgate"], precisions)
        _ssm_targets(b["ssm"], precisions)
    if shared_exp:
        fxp_qconfig["blocks"]["norm"] = _norm_targets((enc[l]["norm"] for l in layers), precisions)
    else:
        for l in layers:
            fxp_qconfig["blocks"][l]["norm"] = _norm_targets([enc[l]["norm"]], precisions)
    return fxp_qconfig


def derive(params: dict, stats: dict, quantization: str = "w8a16", separate_exponents: bool = False,
           precisions: Optional[Dict[str, int]] = None) -> Tuple[dict, dict]:
    """The whole of fxprun.py:294-397 in one call: (modeldict, fxp_qconfig) from the two calibration trees."""
    md = load_modeldict(params, stats)
    prec = precisions or precisions_for(quantization)
    qc = create_fxp_qconfig(md, agg=None) if separate_exponents else create_fxp_qconfig(md, agg="max")[1]
    return md, add_target_bits_exp(md, qc, prec, shared_exp=not separate_exponents)
