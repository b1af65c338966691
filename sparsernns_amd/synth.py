"""Synthetic NDNS-shaped S5 models: float ``modeldict`` + ``fxp_qconfig`` in the reference's layouts.

There is no network, dataset or trained checkpoint here, so benchmarks and tests run on
random-init weights of the reference architecture:

* ``modeldict``  follows the tree consumed by the reference model (Appendix B of SURVEY.md;
  ``sparseRNNs/fxpmodel.py:311-312,437-451,853-888,1233,1240,1425``).
* ``fxp_qconfig`` follows ``create_fxp_qconfig(agg="max")`` + ``add_target_bits_exp``
  (``sparseRNNs/fxputils.py:351-401,404-450,453-786``): every exponent is
  ``min(fracbits, bits - 1 - intbits(absmax))`` with ``fracbits`` from a power-of-two
  calibration scale (``sparseRNNs/utils/quantization.py:352-370``) and ``absmax`` observed on a
  float dry run of the same model (the role ``convert.py`` plays for trained checkpoints).

Everything in this file is host-side NumPy that runs once per model; nothing here is on the
hot path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

F32 = np.float32

NDNS_D_IN = 257  # sparseRNNs/dataloaders/dataloading.py:132-134
NDNS_STFT_MEAN = 0.0007  # sparseRNNs/fxprun.py:65

W8A16 = dict(non_ssm_w=8, non_ssm_b=16, non_ssm_act=16, ssm_w=8, ssm_act=16)  # fxprun.py:302-308
W8A8 = dict(non_ssm_w=8, non_ssm_b=8, non_ssm_act=8, ssm_w=8, ssm_act=8)
# not a reference recipe.  full_range: exp = fracbits of the calibrated power-of-two scale instead of
# min(fracbits, bits - 1 - intbits): the reference's rule never spends a fractional bit beyond bits - 1, which is harmless
# at 16 bits and fatal at 4 (a B-bar tensor with absmax 0.015 would quantise to all zeros at exp 3)
W4A8 = dict(non_ssm_w=4, non_ssm_b=8, non_ssm_act=8, ssm_w=4, ssm_act=8, full_range=True)
PRECISIONS = {"w8a16": W8A16, "w8a8": W8A8, "w4a8": W4A8}


def ndns_dims(dim_scale: float = 0.5) -> Dict[str, int]:
    """recipes/ndns.json + main.py:480-485 + train.py:97-101 (conj_sym halves the state)."""
    blocks0, d_model0, ssm0 = 16, 192, 256
    blocks = int(blocks0 * dim_scale)
    H = int(blocks * (d_model0 / blocks0))
    ssm_base = int(blocks * (ssm0 / blocks0))
    return dict(H=H, P=ssm_base // 2, blocks=blocks, block_size=ssm_base // blocks, n_layers=3, d_in=NDNS_D_IN,
                d_out=NDNS_D_IN)


# --------------------------------------------------------------------------------------
# HiPPO-LegS initialisation (model/ssm_init.py:8-75), NumPy
# --------------------------------------------------------------------------------------
def hippo_dplr(N: int):
    q = np.sqrt(1 + 2 * np.arange(N, dtype=np.float64))
    A = -(np.tril(q[:, None] * q[None, :]) - np.diag(np.arange(N, dtype=np.float64)))
    p = np.sqrt(np.arange(N, dtype=np.float64) + 0.5)
    S = A + p[:, None] * p[None, :]
    lam_re = np.mean(np.diagonal(S)) * np.ones(N)
    lam_im, V = np.linalg.eigh(S * -1j)
    return lam_re + 1j * lam_im, V


def _lecun(rng: np.random.Generator, shape, fan_in: int) -> np.ndarray:
    return (rng.standard_normal(shape) / math.sqrt(fan_in)).astype(F32)


def make_float_params(dims: Dict[str, int], seed: int = 1919, bn_scale_bias: bool = False) -> dict:
    """Random-init float parameters in the ``modeldict`` tree (observers are added later)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, P, d_in, d_out = dims["H"], dims["P"], dims["d_in"], dims["d_out"]
    bs, blocks = dims["block_size"], dims["blocks"]
    lam, V = hippo_dplr(bs)
    half = bs // 2
    lam, V = lam[:half], V[:, :half]
    Lambda = np.tile(lam, blocks)  # (P,)
    Vb = np.zeros((bs * blocks, P), dtype=np.complex128)
    for b in range(blocks):
        Vb[b * bs:(b + 1) * bs, b * half:(b + 1) * half] = V
    Vinv = Vb.conj().T  # (P, 2P)

    def dense(k, m):
        return dict(kernel=_lecun(rng, (k, m), k), bias=(0.01 * rng.standard_normal(m)).astype(F32))

    enc = dict(encoder=dense(d_in, H))
    for i in range(dims["n_layers"]):
        Bm = _lecun(rng, (2 * P, H), 2 * P)
        VinvB = Vinv @ Bm
        Cm = _lecun(rng, (H, 2 * P, 2), 2 * P)
        CV = (Cm[..., 0] + 1j * Cm[..., 1]) @ Vb
        norm = dict(mean=(0.1 * rng.standard_normal(H)).astype(F32), var=rng.uniform(0.5, 1.5, H).astype(F32))
        if bn_scale_bias:
            norm["scale"] = rng.uniform(0.5, 1.5, H).astype(F32) * rng.choice([-1.0, 1.0], H, p=[0.1, 0.9]).astype(F32)
            norm["bias"] = (0.1 * rng.standard_normal(H)).astype(F32)
        enc[f"layers_{i}"] = dict(
            norm=norm,
            mixer=dict(
                Lambda_re=Lambda.real.astype(F32), Lambda_im=Lambda.imag.astype(F32),
                B=np.stack([VinvB.real, VinvB.imag], -1).astype(F32),
                C=np.stack([CV.real, CV.imag], -1).astype(F32),
                D=rng.standard_normal(H).astype(F32),
                log_step=rng.uniform(math.log(1e-3), math.log(1e-1), (P, 1)).astype(F32)),
            out2=dense(H, H))
    return dict(encoder=enc, decoder=dense(H, d_out))


def make_input(B: int, L: int, d_in: int = NDNS_D_IN, seed: int = 0, scale: float = 1.0) -> np.ndarray:
    """NDNS-STFT-magnitude-like input: Exponential(mean 7e-4) - 7e-4 (fxprun.py:65-68)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return ((rng.exponential(NDNS_STFT_MEAN, (B, L, d_in)) - NDNS_STFT_MEAN) * scale).astype(F32)


def prune_magnitude(modeldict: dict, sparsity: float) -> dict:
    """Unstructured per-tensor magnitude mask on every >=2-D weight (kernels, B, C), zeros kept
    densely -- what ``iterative-ste-mag-*`` leaves behind (utils/pruning.py:22-54)."""

    def mask(w):
        mag = np.abs(w) if w.ndim == 2 else np.hypot(w[..., 0], w[..., 1])
        k = int(round(sparsity * mag.size))
        if k <= 0:
            return w
        thr = np.partition(mag.ravel(), k - 1)[k - 1]
        keep = mag > thr
        return (w * (keep if w.ndim == 2 else keep[..., None])).astype(F32)

    enc = modeldict["encoder"]
    enc["encoder"]["kernel"] = mask(enc["encoder"]["kernel"])
    modeldict["decoder"]["kernel"] = mask(modeldict["decoder"]["kernel"])
    for k, layer in enc.items():
        if k.startswith("layers_"):
            layer["out2"]["kernel"] = mask(layer["out2"]["kernel"])
            layer["mixer"]["B"] = mask(layer["mixer"]["B"])
            layer["mixer"]["C"] = mask(layer["mixer"]["C"])
    return modeldict


# --------------------------------------------------------------------------------------
# float dry run (calibration).  Float semantics of the model the fxp path approximates
# (model/ssm.py:37-50,84-185; model/layers.py GLU "half1", prenorm BN, relufication).
# --------------------------------------------------------------------------------------
def zoh(mixer: dict):
    lam = (mixer["Lambda_re"] + 1j * mixer["Lambda_im"]).astype(np.complex64)
    step = np.exp(mixer["log_step"][:, 0]).astype(F32)
    lam_bar = np.exp(lam * step).astype(np.complex64)
    Bt = (mixer["B"][..., 0] + 1j * mixer["B"][..., 1]).astype(np.complex64)
    B_bar = ((1 / lam * (lam_bar - 1))[:, None] * Bt).astype(np.complex64)
    C = (mixer["C"][..., 0] + 1j * mixer["C"][..., 1]).astype(np.complex64)
    return lam_bar, B_bar, C


def float_forward(modeldict: dict, x: np.ndarray, n_layers: int, calibrate_bn: bool = False,
                  stats: Optional[dict] = None, activations: Optional[dict] = None) -> np.ndarray:
    """x: (B,L,d_in) float32.  Records absmax observers into ``stats`` when given, and -- for the verification report
    (sparsernns_amd/fxpreporter.py) -- the float intermediates of sequence 0 into ``activations`` under the keys the
    reference's float model sows (sparseRNNs/model/layers.py:181-243: input, pre_s5, pre_C, pre_GLU, out2/__call__, the
    layer's __call__; ssm.py: mixer/__call__, B_bar; post_GLU = the second drop/__call__ that fxprun.py:688-698 reads)."""

    def obs(key, v):
        if stats is not None:
            stats[key] = max(stats.get(key, 0.0), float(np.abs(v).max()))

    enc = modeldict["encoder"]
    obs("encoder.inp", x)
    h = x @ enc["encoder"]["kernel"] + enc["encoder"]["bias"]
    obs("encoder.out", h)
    h = np.maximum(h, 0)
    act_enc = activations.setdefault("encoder", {}) if activations is not None else None
    for i in range(n_layers):
        layer = enc[f"layers_{i}"]
        skip = h
        act = act_enc.setdefault(f"layers_{i}", {}) if act_enc is not None else None
        nm = layer["norm"]
        if calibrate_bn:
            nm["mean"] = h.mean(axis=(0, 1)).astype(F32)
            nm["var"] = np.maximum(h.var(axis=(0, 1)), 1e-6).astype(F32)
        u = (h - nm["mean"]) / np.sqrt(nm["var"] + F32(1e-5))
        if "scale" in nm:
            u = u * nm["scale"]
        if "bias" in nm:
            u = u + nm["bias"]
        obs(f"l{i}.u", u)
        lam_bar, B_bar, C = zoh(layer["mixer"])
        Bu = u.astype(np.complex64) @ B_bar.T
        obs(f"l{i}.Bu_re", Bu.real)
        obs(f"l{i}.Bu_im", Bu.imag)
        xs = np.empty_like(Bu)
        st = np.zeros((Bu.shape[0], Bu.shape[2]), dtype=np.complex64)
        for t in range(Bu.shape[1]):
            st = lam_bar * st + Bu[:, t]
            xs[:, t] = st
        obs(f"l{i}.x_re", xs.real)
        obs(f"l{i}.x_im", xs.imag)
        keep = (xs.real > 0) | ((xs.real == 0) & (xs.imag > 0))
        xs = np.where(keep, xs, 0)
        y = 2 * (xs.real @ C.real.T - xs.imag @ C.imag.T) + layer["mixer"]["D"] * u
        obs(f"l{i}.y", y)
        x1 = np.maximum(y, 0)
        obs(f"l{i}.out2.inp", x1)
        g_in = x1 @ layer["out2"]["kernel"] + layer["out2"]["bias"]
        obs(f"l{i}.out2.out", g_in)
        g = 1 / (1 + np.exp(-g_in))
        obs(f"l{i}.gate.l", x1)
        obs(f"l{i}.gate.r", g)
        h = np.maximum(x1 * g + skip, 0).astype(F32)
        if act is not None:
            act.update(input=skip[0].astype(F32), pre_s5=u[0].astype(F32), pre_C=xs[0].astype(np.complex64), pre_GLU=y[0].astype(F32),
                       post_GLU=(x1 * g)[0].astype(F32), __call__=h[0])
            act["mixer"] = dict(B_bar=B_bar.astype(np.complex64), __call__=y[0].astype(F32))
            act["out2"] = dict(__call__=g_in[0].astype(F32))
    obs("decoder.inp", h)
    out = h @ modeldict["decoder"]["kernel"] + modeldict["decoder"]["bias"]
    obs("decoder.out", out)
    if activations is not None:
        activations["__call__"] = out[0].astype(F32)
    return out.astype(F32)


# --------------------------------------------------------------------------------------
# qconfig derivation (fxputils.py rules)
# --------------------------------------------------------------------------------------
def get_intbits(absmax: float) -> int:
    """fxputils.py:137-142 (an exact power of two needs one more integer bit)."""
    l2 = math.log2(absmax)
    return max(0, math.ceil(l2)) + (1 if round(l2) == l2 else 0)


def fracbits_from_absmax(absmax: float, bits: int) -> int:
    """-log2 of the calibrated power-of-two scale (utils/quantization.py:352-370;
    fxputils.py:178,249)."""
    return -int(round(math.log2(absmax / float((1 << (bits - 1)) - 1))))


def _entry(absmax, bits: int, full_range: bool = False) -> dict:
    """absmax: one value, or one per layer.  Shared exponents take the maximum over layers of EVERY field separately
    (fxputils.py:288-348): the largest intbits and the largest fracbits, which need not come from the same layer."""
    vals = [max(float(a), 1e-12) for a in (absmax if isinstance(absmax, (list, tuple)) else [absmax])]
    ib, fb = max(get_intbits(a) for a in vals), max(fracbits_from_absmax(a, bits) for a in vals)
    return dict(absmax=max(vals), intbits=ib, signbits=1, fracbits=fb, bits=bits, exp=fb if full_range else min(fb, bits - 1 - ib))


def derive_qconfig(modeldict: dict, stats: dict, n_layers: int, precisions: dict = W8A16) -> dict:
    """Shared-exponent (``--separate_exponents`` absent) fxp_qconfig, fxputils.py:351-401,453-786."""
    enc = modeldict["encoder"]
    layers = [enc[f"layers_{i}"] for i in range(n_layers)]
    wb, bb, ab = precisions["non_ssm_w"], precisions["non_ssm_b"], precisions["non_ssm_act"]
    sw, sa = precisions["ssm_w"], precisions["ssm_act"]
    full = bool(precisions.get("full_range", False))

    def _entry(absmax, bits):  # noqa: F811 - the recipe's exponent rule
        return globals()["_entry"](absmax, bits, full)

    def dense_cfg(dense_list, inp_absmax, out_absmax):
        w = _entry([np.abs(d["kernel"]).max() for d in dense_list], wb)
        inp, out = _entry(inp_absmax, ab), _entry(out_absmax, ab)
        b = _entry([np.abs(d["bias"]).max() for d in dense_list], bb)
        b["fracbits"] = inp["fracbits"]  # the bias shares the input's scale (act_scale), fxputils.py:266
        b["exp"] = b["fracbits"] if full else min(b["fracbits"], bb - 1 - b["intbits"])
        cfg = {}
        for p, e in (("w", w), ("b", b), ("inp", inp), ("out", out)):
            cfg.update({f"{p}_bits": e["bits"], f"{p}_exp": e["exp"], f"{p}_absmax": e["absmax"],
                        f"{p}_intbits": e["intbits"], f"{p}_fracbits": e["fracbits"], f"{p}_signbit": 1})
        return cfg

    lmax = lambda key: [stats[f"l{i}.{key}"] for i in range(n_layers)]  # one value per layer
    zs = [zoh(l["mixer"]) for l in layers]
    wts = dict(
        A_re=_entry([np.abs(z[0].real).max() for z in zs], sa),  # fxputils.py:557-562: Lambda gets act bits
        A_im=_entry([np.abs(z[0].imag).max() for z in zs], sa),
        B_re=_entry([np.abs(z[1].real).max() for z in zs], sw),
        B_im=_entry([np.abs(z[1].imag).max() for z in zs], sw),
        C_re=_entry([np.abs(z[2].real).max() for z in zs], sw),
        C_im=_entry([np.abs(z[2].imag).max() for z in zs], sw),
        D=_entry([np.abs(l["mixer"]["D"]).max() for l in layers], sw))
    acts = {k: _entry(lmax(k), sa) for k in ["u", "Bu_re", "Bu_im", "x_re", "x_im", "y"]}
    gl, gr = _entry(lmax("gate.l"), ab), _entry(lmax("gate.r"), ab)
    multgate = dict(l_bits=ab, l_exp=gl["exp"], r_bits=ab, r_exp=gr["exp"], res_bits=ab,
                    res_exp=ab - 1 - (gl["intbits"] + gr["intbits"]),  # fxputils.py:530-537
                    l_intbits=gl["intbits"], r_intbits=gr["intbits"], l_absmax=gl["absmax"], r_absmax=gr["absmax"],
                    l_fracbits=gl["fracbits"], r_fracbits=gr["fracbits"])

    def norm_entry(vals):  # fxputils.py:636-752: no power-of-two bump here
        ib = max(0, math.ceil(math.log2(max(float(np.abs(v).max()) for v in vals))))
        return dict(intbits=ib, exp=ab - 1 - ib, bits=ab)

    norm = dict(mean=norm_entry([l["norm"]["mean"] for l in layers]),
                var=norm_entry([l["norm"]["var"] for l in layers]),
                invsq_var=norm_entry([1.0 / np.sqrt(l["norm"]["var"] + F32(1e-5)) for l in layers]))
    for k in ("scale", "bias"):
        if k in layers[0]["norm"]:
            norm[k] = norm_entry([l["norm"][k] for l in layers])
    return dict(
        encoder=dense_cfg([enc["encoder"]], stats["encoder.inp"], stats["encoder.out"]),
        blocks=dict(ssm=dict(weights=wts, activations=acts), multgate=multgate,
                    out2=dense_cfg([l["out2"] for l in layers], lmax("out2.inp"), lmax("out2.out")), norm=norm),
        decoder=dense_cfg([modeldict["decoder"]], stats["decoder.inp"], stats["decoder.out"]))


def make_model(dim_scale: float = 0.5, seed: int = 1919, quantization: str = "w8a16", sparsity: float = 0.0,
               calib_B: int = 2, calib_L: int = 256, input_scale: float = 1.0, bn_stats: str = "calibrated",
               bn_scale_bias: bool = False, dims: Optional[dict] = None,
               state_headroom_bits: int = 0) -> Tuple[dict, dict, dict]:
    """Returns (modeldict, fxp_qconfig, dims).

    bn_stats="calibrated": BatchNorm running mean/var are set to the statistics of the layer
    input on the calibration batch (what training would leave behind), so activations use the
    full 16-bit range; "random": mean ~ N(0,0.1), var ~ U(0.5,1.5) as drawn.

    state_headroom_bits: extra integer bits for the SSM state (x_re/x_im exponents lowered by that
    much).  The reference never clips the state (fxpmodel.py:147-172), and the fixed-point model
    drifts from the float calibration run through its saturation quirks, so a float-calibrated state
    exponent can be exceeded on long sequences; a deployment would calibrate with this margin.
    """
    dims = dict(dims) if dims is not None else ndns_dims(dim_scale)
    md = make_float_params(dims, seed, bn_scale_bias)
    if sparsity > 0:
        md = prune_magnitude(md, sparsity)
    stats: dict = {}
    xcal = make_input(calib_B, calib_L, dims["d_in"], seed=seed + 1, scale=input_scale)
    float_forward(md, xcal, dims["n_layers"], calibrate_bn=(bn_stats == "calibrated"), stats=stats)
    qc = derive_qconfig(md, stats, dims["n_layers"], PRECISIONS[quantization])
    for k in ("x_re", "x_im"):
        qc["blocks"]["ssm"]["activations"][k]["exp"] -= state_headroom_bits
    cap_result_exponents(qc)
    _assert_exps_nonnegative(qc)
    return md, qc, dims


def cap_result_exponents(qc: dict) -> dict:
    """Lowers a result exponent that asks for more fractional bits than its operands carry.

    Every product in the model is shifted right by ``e1 + e2 - result_exp`` (fxparray.py:619-621, 662-664); a negative
    shift is a ``ValueError`` in ``fxp_mul`` and undefined in ``fxp_matmul``.  ``add_target_bits_exp`` never produces
    one for the reference's 16-bit activation recipes, but with 8-bit activations and 4-bit weights (w4a8,
    SURVEY.md 8d C5) ``min(fracbits, bits-1-intbits)`` of a small product can exceed what a 4-bit kernel and an
    8-bit input provide.  Capping the result exponent at the sum of the operand exponents (shift 0: the product is
    kept exactly) is the calibration a narrow recipe needs; it leaves every w8a16 configuration untouched.
    """
    def cap(d, key, limit):
        if d[key] > limit:
            d[key] = limit

    for name in ("encoder", "decoder"):
        cap(qc[name], "out_exp", qc[name]["inp_exp"] + qc[name]["w_exp"])
    blocks = [qc["blocks"]] + [v for k, v in qc["blocks"].items() if k.startswith("layers_")]
    for b in blocks:
        if "ssm" not in b:
            continue
        w, a = b["ssm"]["weights"], b["ssm"]["activations"]
        cap(a["Bu_re"], "exp", a["u"]["exp"] + w["B_re"]["exp"])
        cap(a["Bu_im"], "exp", a["u"]["exp"] + w["B_im"]["exp"])
        cap(a["y"], "exp", min(a["x_re"]["exp"] + w["C_re"]["exp"], a["x_im"]["exp"] + w["C_im"]["exp"],
                               w["D"]["exp"] + a["u"]["exp"]))
        # a dense layer converts its input only when that is finer than inp_exp (fxpmodel.py:335-347)
        cap(b["out2"], "out_exp", min(b["out2"]["inp_exp"], a["y"]["exp"]) + b["out2"]["w_exp"])
        cap(b["multgate"], "res_exp", b["multgate"]["l_exp"] + b["multgate"]["r_exp"])
    return qc


def _assert_exps_nonnegative(tree, path="fxp_qconfig"):
    """A negative exponent is not representable in the reference (``1 << exp``, fxparray.py:73)."""
    for k, v in tree.items():
        if isinstance(v, dict):
            _assert_exps_nonnegative(v, f"{path}.{k}")
        elif (k == "exp" or k.endswith("_exp")) and v < 0:
            raise ValueError(f"{path}.{k} = {v} < 0: this synthetic configuration does not fit the chosen bit widths "
                             "(try bn_stats='random' or a larger input_scale)")


def tiny_dims(H: int = 8, P: int = 4, d_in: int = 5, d_out: int = 5, n_layers: int = 2) -> dict:
    """Small shapes for fixtures: P must be blocks * block_size/2 with an even block_size."""
    return dict(H=H, P=P, blocks=P // 2, block_size=4, n_layers=n_layers, d_in=d_in, d_out=d_out)


# --------------------------------------------------------------------------------------
# the same calibration result in the reference's checkpoint layout (what convert.py leaves behind)
# --------------------------------------------------------------------------------------
def reference_trees(modeldict: dict, stats: dict, n_layers: int, precisions: dict = W8A16) -> Tuple[dict, dict]:
    """(params, stats) trees shaped like ``sc_calibrated_params.pkl`` / ``sc_cal_stats.pkl``:
    float parameters on one side; on the other, per observed tensor, the calibrated power-of-two ``scale``
    (utils/quantization.py:352-370) and an ``observer`` with its min / max.  ``sparsernns_amd.fxputils.derive`` turns
    them back into (modeldict, fxp_qconfig); the CPU suite checks that this reproduces ``derive_qconfig``."""
    wb, ab = precisions["non_ssm_w"], precisions["non_ssm_act"]
    sw, sa = precisions["ssm_w"], precisions["ssm_act"]

    def pow2_scale(absmax, bits):
        return F32(2.0 ** round(math.log2(max(float(absmax), 1e-12) / float((1 << (bits - 1)) - 1))))

    def observed(absmax, bits, stem=""):
        a = F32(max(float(absmax), 1e-12))
        return dict(scale=pow2_scale(a, bits), observer={f"{stem}observer_min": -a, f"{stem}observer_max": a})

    def dense_stats(d, inp_absmax, out_absmax):
        a_in, a_out = F32(max(float(inp_absmax), 1e-12)), F32(max(float(out_absmax), 1e-12))
        return dict(act_scale=pow2_scale(a_in, ab), weight_scale=pow2_scale(np.abs(d["kernel"]).max(), wb),
                    out_scale=pow2_scale(a_out, ab),
                    input_observer=dict(input_observer_min=-a_in, input_observer_max=a_in),
                    output_observer=dict(output_observer_min=-a_out, output_observer_max=a_out))

    enc = modeldict["encoder"]
    params = dict(encoder=dict(encoder=dict(enc["encoder"])), decoder=dict(modeldict["decoder"]))
    st = dict(encoder=dict(encoder=dense_stats(enc["encoder"], stats["encoder.inp"], stats["encoder.out"])),
              decoder=dense_stats(modeldict["decoder"], stats["decoder.inp"], stats["decoder.out"]))
    for i in range(n_layers):
        layer = enc[f"layers_{i}"]
        lam_bar, B_bar, C = zoh(layer["mixer"])
        cplx = lambda z, bits: dict(quant_real=observed(np.abs(z.real).max(), bits), quant_imag=observed(np.abs(z.imag).max(), bits))
        act2 = lambda k: dict(quant_real=observed(stats[f"l{i}.{k}_re"], sa), quant_imag=observed(stats[f"l{i}.{k}_im"], sa))
        params["encoder"][f"layers_{i}"] = dict(norm=dict(layer["norm"]), mixer=dict(layer["mixer"]), out2=dict(layer["out2"]))
        st["encoder"][f"layers_{i}"] = dict(
            mixer=dict(quant_A=cplx(lam_bar, sa), quant_B=cplx(B_bar, sw), quant_C=cplx(C, sw),
                       quant_D=observed(np.abs(layer["mixer"]["D"]).max(), sw), quant_ut=observed(stats[f"l{i}.u"], sa),
                       quant_But=act2("Bu"), quant_xt=act2("x"), quant_yt=observed(stats[f"l{i}.y"], sa)),
            mult_gate=dict(quant_left=observed(stats[f"l{i}.gate.l"], ab), quant_right=observed(stats[f"l{i}.gate.r"], ab)),
            out2=dense_stats(layer["out2"], stats[f"l{i}.out2.inp"], stats[f"l{i}.out2.out"]))
    return params, st
