"""Fixed-point S5 model on MI355X: host-side mirror of the reference's ``sparseRNNs/fxpmodel.py``.

Same classes, constructor keywords, ``export()`` layout and ``intermediates`` names as the
reference, so ``fxprun.py``-style harness code runs unchanged against it:

    model = FxpRegressionModel(modeldict=..., fxp_qconfig=..., scope="model",
                               mixer_cls=FxpSSM.init_fn(H=..., P=..., discretization="zoh"), ...)
    y = model(fxp_x)            # FxpArray in, FxpArray out

Two execution modes, both on the GPU through the C ABI of ``include/s5fxp.h``:

* fused (default, ``store_intermediates=False``): one call of ``s5fxp_model_forward`` --
  data-dependent exponents stay on the device, no host sync inside the forward;
* eager (``store_intermediates=True``): op by op through ``sparsernns_amd.fxparray`` exactly as the
  reference does it, recording every intermediate under the reference's ``sow`` names.

Parameter quantisation (``setup``) is host-side NumPy that runs once per model.
Citations are file:line into /root/reference/sparseRNNs/fxpmodel.py unless a file is named.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict
from dataclasses import dataclass, field
from functools import partial
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np

F32 = np.float32

GLU_VARIANTS = ["full", "half1", "half2", "none"]


# --------------------------------------------------------------------------------------
# host-side quantisation helpers (setup only)
# --------------------------------------------------------------------------------------
def host_from_fp(x, bits: int, exp: int, round_mode: str = "round") -> np.ndarray:
    """fxp_from_fp on the host for parameters (fxparray.py:287-307): float32 scale, round, clip."""
    xi = (np.asarray(x, dtype=F32) * F32(1 << exp)).astype(F32)
    r = np.rint(xi) if round_mode == "round" else (np.floor(xi) if round_mode == "floor" else np.ceil(xi))
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
    return np.clip(r.astype(np.float64), lo, hi).astype(np.int64).astype(np.int32)


def discretize_zoh(Lambda, B_tilde, Delta):
    """model/ssm.py:37-50 in complex64."""
    Lambda = np.asarray(Lambda, dtype=np.complex64)
    Delta = np.asarray(Delta, dtype=F32)
    Lambda_bar = np.exp((Lambda * Delta).astype(np.complex64)).astype(np.complex64)
    ident = np.ones(Lambda.shape[0], dtype=F32)
    B_bar = (np.complex64(1) / Lambda * (Lambda_bar - ident)).astype(np.complex64)[..., None] * np.asarray(
        B_tilde, dtype=np.complex64)
    return Lambda_bar, B_bar.astype(np.complex64)


def sigmoid_lut(x_exp: int, y_exp: int, x_extra: int = 3, n_exp: int = 3) -> np.ndarray:
    """:89-95 (``1 << a + b`` is ``1 << (a + b)``): lut[k] = rint(sigmoid(k) * 2^y_exp) - 2^(y_exp-1)."""
    x = np.linspace(0, 1 << (x_exp + x_extra), (1 << n_exp) + 1, dtype=F32)[:-1].astype(np.int32)
    xf = (x.astype(F32) / F32(1 << x_exp)).astype(F32)
    s = (F32(1) / (F32(1) + np.exp(-xf).astype(F32))).astype(F32)
    return (np.rint((s * F32(1 << y_exp)).astype(F32)) - F32(1 << (y_exp - 1))).astype(np.int32)


class QuantizationConfig:
    """Stand-in for utils/quantization.py's QuantizationConfig: the fxp path only type-checks it
    (:221,244-246)."""

    @staticmethod
    def none():
        return QuantizationConfig()

    def to_dict(self):
        return {}


@dataclass(kw_only=True)
class FxpS5Config:  # :211-256
    H: int
    P: int
    discretization: str
    conj_sym: bool = True
    bidirectional: bool = False
    associative_scan: bool = False
    q_config: Any = None
    n_layers: int
    d_model: int
    batchnorm: bool = True
    prenorm: bool = False
    bn_momentum: float = 0.9
    glu_variant: str = "none"
    step_rescale: float = 1.0
    relufication: bool = False
    fuse_batchnorm_linear: bool = False
    dropout: float = 0.2
    training: bool = True
    d_output: int = None
    padded: bool = False

    def __post_init__(self):
        assert self.discretization in ["zoh", "foh"], f"Invalid discretization: {self.discretization}"
        assert self.glu_variant in GLU_VARIANTS, f"Invalid GLU variant: {self.glu_variant}"


# --------------------------------------------------------------------------------------
# device ops are imported lazily so that model SETUP (pure host code) works without the .so
# --------------------------------------------------------------------------------------
def _fx():
    from . import fxparray
    return fxparray


def fxp_relu(x):
    """:27-63."""
    fx = _fx()
    import torch
    from ._lib import check, lib
    if isinstance(x, fx.ComplexFxpArray):
        re, im = x.real.data.contiguous(), x.imag.data.contiguous()
        ore, oim = torch.empty_like(re), torch.empty_like(im)
        check(lib.s5fxp_relu(re.data_ptr(), im.data_ptr(), ore.data_ptr(), oim.data_ptr(), re.numel(),
                             torch.cuda.current_stream().cuda_stream), "fxp_relu")
        return fx.ComplexFxpArray(fx.FxpArray(ore, x.real.bits, x.real.exp, x.real.signed),
                                  fx.FxpArray(oim, x.imag.bits, x.imag.exp, x.imag.signed))
    re = x.data.contiguous()
    ore = torch.empty_like(re)
    check(lib.s5fxp_relu(re.data_ptr(), None, ore.data_ptr(), None, re.numel(),
                         torch.cuda.current_stream().cuda_stream), "fxp_relu")
    return fx.FxpArray(ore, x.bits, x.exp, x.signed)


class FxpSigmoid:
    """:70-144."""

    def __init__(self, x_exp: int = 6, y_exp: int = 8, x_extra: int = 3, n_exp: int = 3):
        if x_extra != 3 or n_exp != 3:
            raise NotImplementedError("only the reference's 8-entry LUT (x_extra=3, n_exp=3) is implemented")
        self.x_exp, self.y_exp, self.x_extra, self.n_exp = x_exp, y_exp, x_extra, n_exp
        self.lut = sigmoid_lut(x_exp, y_exp, x_extra, n_exp)

    def apply(self, x, output_fxp: bool = True):
        fx = _fx()
        import torch
        from ._lib import check, lib
        if not isinstance(x, fx.FxpArray):
            raise NotImplementedError("FxpSigmoid.apply: only FxpArray inputs are implemented")
        src = x.data.contiguous()
        y = torch.empty_like(src)
        lut = (C.c_int32 * 8)(*[int(v) for v in self.lut])
        check(lib.s5fxp_sigmoid(src.data_ptr(), y.data_ptr(), src.numel(), x.bits, x.exp, self.x_exp, self.y_exp, lut,
                                torch.cuda.current_stream().cuda_stream), "FxpSigmoid.apply")
        out = fx.FxpArray(y, x.bits, self.y_exp, True)
        return out if output_fxp else out.to_float()


@dataclass
class FxpModule:  # :259-288
    modeldict: Any
    fxp_qconfig: Any
    scope: str
    store_intermediates: bool

    def setup(self):
        pass

    def forward(self, *args, **kwargs):
        pass

    def sow(self, top_key: str, key: str, value):
        if self.store_intermediates:
            assert top_key == "intermediates", f"Invalid top_key for sow: {top_key}"
            self.intermediates[key].append(value.copy() if hasattr(value, "copy") else value)

    def __post_init__(self):
        self.intermediates = defaultdict(list)
        self.setup()

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def last_intermediates(self):
        return {k: v[-1] if len(v) > 0 else None for k, v in self.intermediates.items()}

    def export(self):
        pass


@dataclass
class HostFxp:
    """A quantised parameter on the host (numpy int32) with its config; `.dev()` -> FxpArray."""

    data: np.ndarray
    bits: int
    exp: int
    signed: bool = True
    _dev: Any = None

    def dev(self):
        if self._dev is None:
            self._dev = _fx().FxpArray(np.ascontiguousarray(self.data), self.bits, self.exp, self.signed)
        return self._dev

    def to_float(self):
        return self.data.astype(F32) / F32(1 << self.exp)

    def transpose(self):
        return HostFxp(np.ascontiguousarray(self.data.T), self.bits, self.exp, self.signed)


@dataclass
class HostComplexFxp:
    real: HostFxp
    imag: HostFxp


CSR_DENSITY = 0.25  # kernels at or below this density use s5fxp_dense_csr in op-by-op mode


@dataclass
class FxpDense(FxpModule):  # :291-393
    weight: Optional[HostFxp] = None
    bias: Optional[HostFxp] = None
    _csr: Optional[object] = None

    def setup(self):
        q = self.fxp_qconfig
        self.weight_exp, self.weight_bits, self.weight_signed = q["w_exp"], q["w_bits"], True
        self.bias_exp, self.bias_bits, self.bias_signed = q["b_exp"], q["b_bits"], True
        self.inp_bits, self.inp_exp, self.out_bits, self.out_exp = q["inp_bits"], q["inp_exp"], q["out_bits"], q["out_exp"]
        self.weight = HostFxp(host_from_fp(self.modeldict["kernel"], self.weight_bits, self.weight_exp),
                              self.weight_bits, self.weight_exp)
        b = self.modeldict.get("bias", None)
        self.bias = None if b is None else HostFxp(host_from_fp(b, self.bias_bits, self.bias_exp), self.bias_bits,
                                                   self.bias_exp)

    def forward(self, x):
        fx = _fx()
        if (x.bits > self.inp_bits) or (x.exp > self.inp_exp):
            x = x.change_cfg(new_bits=self.inp_bits, new_exp=self.inp_exp, new_signed=True)
        # op-by-op mode only (the fused engine keeps weights in registers): a pruned kernel goes through the CSR op
        if self._csr is None and np.count_nonzero(self.weight.data) <= CSR_DENSITY * self.weight.data.size:
            self._csr = fx.CsrWeight(self.weight.data, self.weight_bits, self.weight_exp)
        if self._csr is not None:
            wx = fx.fxp_matmul_csr(x, self._csr, result_bits=self.out_bits, result_exp=self.out_exp)
        else:
            wx = fx.fxp_matmul(x, self.weight.dev(), result_bits=self.out_bits, result_exp=self.out_exp)
        if self.bias is not None:
            wx = fx.fxp_add(wx, self.bias.dev(), result_bits=self.out_bits, result_exp=self.out_exp)
        self.sow("intermediates", "__call__", wx)
        return wx

    def export(self):
        return dict(
            params=dict(weight=self.weight.data, bias=self.bias.data),
            qconfig=dict(weight_exp=self.weight_exp, weight_bits=self.weight_bits, weight_signed=self.weight_signed,
                         bias_exp=self.bias_exp, bias_bits=self.bias_bits, bias_signed=self.bias_signed,
                         inp_bits=self.inp_bits, inp_exp=self.inp_exp, out_bits=self.out_bits, out_exp=self.out_exp),
            intermediates=self.last_intermediates())


@dataclass
class FxpSSM(FxpModule):  # :396-847
    H: int
    P: int
    discretization: str
    conj_sym: bool = True
    q_config: Any = None
    step_rescale: float = 1.0
    bidirectional: bool = False
    relufication: bool = True
    associative_scan: bool = False
    clip_eigs: bool = False
    bn_mean: Any = None
    bn_var: Any = None
    bn_scale: Any = None
    bn_bias: Any = None
    bn_eps: float = 1e-5
    use_lax_scan: bool = True
    compute_fp32: bool = False

    def setup(self):
        assert self.relufication, "Only relufication=True is supported for now"
        assert not self.associative_scan, "Only associative_scan=False is supported for now"
        assert not self.bidirectional, "Only bidirectional=False is supported for now"
        assert not self.clip_eigs, "Only clip_eigs=False is supported for now"
        if self.bn_mean is not None and self.bn_var is not None:
            # the reference's fused-BatchNorm branch cannot run (:537-549 uses a name before it is bound)
            raise NotImplementedError("fuse_batchnorm_linear is broken in the reference and not implemented")
        if self.discretization != "zoh":
            raise NotImplementedError(f"Discretization method {self.discretization}")
        if not self.conj_sym:
            raise NotImplementedError("conj_sym=False is not used by fxprun (fxprun.py:408)")
        md, w = self.modeldict, self.fxp_qconfig["weights"]
        B_tilde = md["B"][..., 0] + 1j * md["B"][..., 1]
        Lambda = md["Lambda_re"] + 1j * md["Lambda_im"]
        step = (F32(self.step_rescale) * np.exp(np.asarray(md["log_step"], dtype=F32)[:, 0])).astype(F32)
        Lambda_bar, B_bar = discretize_zoh(Lambda, B_tilde, step)
        C_tilde = (md["C"][..., 0] + 1j * md["C"][..., 1]).astype(np.complex64)
        q = lambda v, k: HostFxp(host_from_fp(v, w[k]["bits"], w[k]["exp"]), w[k]["bits"], w[k]["exp"])
        self.Lambda_bar = HostComplexFxp(q(Lambda_bar.real, "A_re"), q(Lambda_bar.imag, "A_im"))
        self.B_bar = HostComplexFxp(q(B_bar.real, "B_re"), q(B_bar.imag, "B_im"))
        self.C_tilde = HostComplexFxp(q(C_tilde.real, "C_re"), q(C_tilde.imag, "C_im"))
        self.D = q(md["D"], "D")
        self.B_bias = None
        self.D_bias = None

    def forward(self, input_sequence):
        """Eager, op-by-op (:610-794); the fused path lives in FxpRegressionModel."""
        fx = _fx()
        import torch
        from ._lib import check, lib
        act = self.fxp_qconfig["activations"]
        u = input_sequence.change_cfg(new_bits=act["u"]["bits"], new_exp=act["u"]["exp"], new_signed=True)
        Bu = fx.ComplexFxpArray(
            real=fx.fxp_matmul(u, self.B_bar.real.transpose().dev(), result_exp=act["Bu_re"]["exp"],
                               result_bits=act["Bu_re"]["bits"]),
            imag=fx.fxp_matmul(u, self.B_bar.imag.transpose().dev(), result_exp=act["Bu_im"]["exp"],
                               result_bits=act["Bu_im"]["bits"]))
        self.sow("intermediates", "Bu_elements", Bu)
        bre, bim = Bu.real.data.contiguous(), Bu.imag.data.contiguous()
        B = 1 if bre.ndim == 2 else bre.shape[0]
        L, P = bre.shape[-2], bre.shape[-1]
        xr, xi = torch.empty_like(bre), torch.empty_like(bim)
        A = self.Lambda_bar
        check(lib.s5fxp_scan(bre.data_ptr(), bim.data_ptr(), A.real.dev().data.data_ptr(), A.imag.dev().data.data_ptr(),
                             xr.data_ptr(), xi.data_ptr(), B, L, P, A.real.exp, A.imag.exp, Bu.real.exp, Bu.imag.exp,
                             act["x_re"]["exp"], act["x_im"]["exp"], 0, torch.cuda.current_stream().cuda_stream),
              "recurrent_loop")
        xs = fx.ComplexFxpArray(fx.FxpArray(xr, act["x_re"]["bits"], act["x_re"]["exp"], True),
                                fx.FxpArray(xi, act["x_im"]["bits"], act["x_im"]["exp"], True))
        self.sow("intermediates", "xs", xs)
        xs = fxp_relu(xs)
        self.sow("intermediates", "xs_relu", xs)
        yb, ye = act["y"]["bits"], act["y"]["exp"]
        ys = fx.fxp_sub(fx.fxp_matmul(xs.real, self.C_tilde.real.transpose().dev(), result_exp=ye, result_bits=yb),
                        fx.fxp_matmul(xs.imag, self.C_tilde.imag.transpose().dev(), result_exp=ye, result_bits=yb),
                        result_exp=ye, result_bits=yb)
        self.sow("intermediates", "Cxs", ys)
        if self.conj_sym:
            ys = fx.FxpArray(ys.data * 2, ys.bits, ys.exp, ys.signed)  # int32 wrap, no clip (:765-767)
            self.sow("intermediates", "2Cxs", ys)
        Du = fx.fxp_mul(self.D.dev(), u, result_exp=ye, result_bits=yb)
        self.sow("intermediates", "Du", Du)
        ysDu = fx.fxp_add(ys, Du, result_exp=ye, result_bits=yb)
        self.sow("intermediates", "ys", ysDu)
        return ysDu, xs

    @staticmethod
    def init_fn(H: int, P: int, discretization: str, conj_sym: bool = True, q_config: Any = None,
                bidirectional: bool = False, relufication: bool = True, associative_scan: bool = False):
        return partial(FxpSSM, H=H, P=P, discretization=discretization, conj_sym=conj_sym, q_config=q_config,
                       bidirectional=bidirectional, relufication=relufication, associative_scan=associative_scan)

    def export(self):
        params, qconfig = {}, {}
        d = dict(A_real=self.Lambda_bar.real, A_imag=self.Lambda_bar.imag, B_real=self.B_bar.real,
                 B_imag=self.B_bar.imag, C_real=self.C_tilde.real, C_imag=self.C_tilde.imag, D=self.D)
        for k, x in d.items():
            params[k] = x.data
            qconfig[f"{k}_bits"], qconfig[f"{k}_exp"], qconfig[f"{k}_signed"] = x.bits, x.exp, x.signed
        for k in ["u", "Bu_re", "Bu_im", "x_re", "x_im", "y"]:
            qconfig[f"{k}_bits"] = self.fxp_qconfig["activations"][k]["bits"]
            qconfig[f"{k}_exp"] = self.fxp_qconfig["activations"][k]["exp"]
        return dict(params=params, qconfig=qconfig, intermediates=self.last_intermediates())


@dataclass
class FxpBatchNorm(FxpModule):  # :850-968
    def setup(self, bn_eps: float = 1e-5):
        md, qc = self.modeldict, self.fxp_qconfig
        q = lambda v, k: HostFxp(host_from_fp(v, qc[k]["bits"], qc[k]["exp"]), qc[k]["bits"], qc[k]["exp"])
        self.minus_mean = q(F32(-1) * np.asarray(md["mean"], dtype=F32), "mean")
        self.invsq_var = q(F32(1.0) / np.sqrt(np.asarray(md["var"], dtype=F32) + F32(bn_eps)), "invsq_var")
        self.bias = q(md["bias"], "bias") if "bias" in md else None
        self.scale = q(md["scale"], "scale") if "scale" in md else None

    def forward(self, x):
        fx = _fx()
        self.sow("intermediates", "norm_input", x)
        x = fx.fxp_add(x, self.minus_mean.dev(), result_exp="compute_best")
        self.sow("intermediates", "norm_input_minus_mean", x)
        x = fx.fxp_mul(x, self.invsq_var.dev(), result_exp="compute_best")
        self.sow("intermediates", "norm_output_raw", x)
        if self.scale is not None:
            x = fx.fxp_mul(x, self.scale.dev(), result_exp="compute_best")
            self.sow("intermediates", "norm_output_scaled", x)
        if self.bias is not None:
            x = fx.fxp_add(x, self.bias.dev(), result_exp="compute_best")
            self.sow("intermediates", "norm_output_scaled_bias", x)
        self.sow("intermediates", "norm_output", x)
        return x

    def export(self):
        data = dict(
            params=dict(mean=(-1 * self.minus_mean.data.astype(np.int64)).astype(np.int32),
                        invsq_var=self.invsq_var.data),
            qconfig=dict(mean_bits=self.minus_mean.bits, mean_exp=self.minus_mean.exp,
                         invsq_var_bits=self.invsq_var.bits, invsq_var_exp=self.invsq_var.exp),
            intermediates=self.last_intermediates())
        if self.bias is not None:
            data["params"]["bias"] = self.bias.data
            data["qconfig"]["bias_bits"], data["qconfig"]["bias_exp"] = self.bias.bits, self.bias.exp
        if self.scale is not None:
            data["params"]["scale"] = self.scale.data
            data["qconfig"]["scale_bits"], data["qconfig"]["scale_exp"] = self.scale.bits, self.scale.exp
        return data


@dataclass
class FxpSequenceLayer(FxpModule):  # :971-1207
    mixer_cls: Any
    d_model: int
    batchnorm: bool = True
    prenorm: bool = True
    glu_variant: str = "none"
    bn_momentum: float = 0.90
    step_rescale: float = 1.0
    relufication: bool = False
    fuse_batchnorm_linear: bool = False
    q_config: Any = None
    dropout: float = 0.2
    training: bool = True
    layer_idx: Optional[int] = None

    def setup(self):
        keys = [e for e in self.fxp_qconfig.keys() if e.startswith("layers_")]
        if len(keys) > 0:
            self.fxp_qconfig = self.fxp_qconfig[f"layers_{self.layer_idx}"]
        assert self.batchnorm, "Only batchnorm is supported for now."
        assert self.dropout == 0.0, "Only dropout=0.0 is supported for now."
        assert not self.training, "Only training=False is supported for now."
        assert self.relufication, "Only relufication=True is supported for now."
        assert self.prenorm, "Only prenorm=True is supported for now."
        if self.fuse_batchnorm_linear:
            raise NotImplementedError("fuse_batchnorm_linear is broken in the reference (fxpmodel.py:537-549) "
                                      "and not implemented")
        assert self.glu_variant in GLU_VARIANTS, f"GLU variant must be one of {GLU_VARIANTS}"
        if self.glu_variant != "half1":
            raise NotImplementedError("only glu_variant='half1' (the NDNS recipe, recipes/ndns.json) is implemented")
        self.norm = FxpBatchNorm(modeldict=self.modeldict["norm"], fxp_qconfig=self.fxp_qconfig["norm"],
                                 scope=f"{self.scope}.norm", store_intermediates=self.store_intermediates)
        self.mixer = self.mixer_cls(
            modeldict=self.modeldict["seq"] if "seq" in self.modeldict else self.modeldict["mixer"],
            fxp_qconfig=self.fxp_qconfig["ssm"], scope=f"{self.scope}.mixer",
            store_intermediates=self.store_intermediates, step_rescale=self.step_rescale)
        self.out2 = FxpDense(modeldict=self.modeldict["out2"], fxp_qconfig=self.fxp_qconfig["out2"],
                             scope=f"{self.scope}.out2", store_intermediates=self.store_intermediates)
        self.glu_act_fn = fxp_relu
        sigmoid_x_exp = min(self.fxp_qconfig["out2"]["out_exp"], 6)  # :1097-1103
        sigmoid_y_exp = self.fxp_qconfig["out2"]["out_bits"] - 2
        self.sigmoid_cls = FxpSigmoid(x_exp=sigmoid_x_exp, y_exp=sigmoid_y_exp)
        self.sigmoid = partial(self.sigmoid_cls.apply, output_fxp=True)

    def mult_gate(self, x, y):  # :1075-1093
        mg = self.fxp_qconfig["multgate"]
        x = x.change_cfg(new_bits=mg["l_bits"], new_exp=mg["l_exp"], new_signed=True)
        y = y.change_cfg(new_bits=mg["r_bits"], new_exp=mg["r_exp"], new_signed=True)
        return _fx().fxp_mul(x, y, result_exp=mg["res_exp"], result_bits=mg["res_bits"])

    def forward(self, x):
        fx = _fx()
        skip = x
        self.sow("intermediates", "ssm_input", x)
        x = self.norm(x)
        self.sow("intermediates", "pre_s5", x)
        x, x_pre_C = self.mixer(x)
        self.sow("intermediates", "pre_C", x_pre_C)
        x1 = self.glu_act_fn(x)
        self.sow("intermediates", "pre_GLU", x)
        rside = self.sigmoid(self.out2(x1))
        self.sow("intermediates", "out2_sigmoid", rside)
        x = self.mult_gate(x1, rside)
        self.sow("intermediates", "post_GLU", x)
        x = fx.fxp_add(x, skip, result_exp="compute_best", result_bits=self.fxp_qconfig["multgate"]["res_bits"])
        self.sow("intermediates", "residadd", x)
        x = fxp_relu(x)
        self.sow("intermediates", "output", x)
        return x

    def export(self):
        norm_data, mixer_data, out2_data = self.norm.export(), self.mixer.export(), self.out2.export()
        mg = self.fxp_qconfig["multgate"]
        multgate_qconfig = {k: mg[k] for k in ("l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp")}
        sigmoid_qconfig = dict(x_exp=self.sigmoid_cls.x_exp, y_exp=self.sigmoid_cls.y_exp,
                               x_extra=self.sigmoid_cls.x_extra, n_exp=self.sigmoid_cls.n_exp)
        return dict(
            params=dict(mixer=mixer_data["params"], out2=out2_data["params"], norm=norm_data["params"]),
            qconfig=dict(mixer=mixer_data["qconfig"], out2=out2_data["qconfig"], multgate=multgate_qconfig,
                         sigmoid=sigmoid_qconfig, norm=norm_data["qconfig"]),
            intermediates=dict(mixer=mixer_data["intermediates"], out2=out2_data["intermediates"],
                               norm=norm_data["intermediates"], **self.last_intermediates()))


_LAYER_KW = ("d_model", "dropout", "batchnorm", "prenorm", "glu_variant", "bn_momentum", "training", "step_rescale",
             "relufication", "fuse_batchnorm_linear", "q_config")


@dataclass
class FxpStackedEncoderModel(FxpModule):  # :1210-1289
    mixer_cls: Any
    n_layers: int
    d_model: int
    batchnorm: bool = True
    prenorm: bool = False
    bn_momentum: float = 0.9
    glu_variant: str = "none"
    step_rescale: float = 1.0
    relufication: bool = False
    fuse_batchnorm_linear: bool = False
    q_config: Any = None
    dropout: float = 0.2
    training: bool = True

    def setup(self):
        assert self.batchnorm, "Only batchnorm is supported for now."
        assert self.dropout == 0.0, "Only dropout=0.0 is supported for now."
        assert not self.training, "Only training=False is supported for now."
        assert self.relufication, "Only relufication=True is supported for now."
        self.encoder = FxpDense(modeldict=self.modeldict["encoder"], fxp_qconfig=self.fxp_qconfig["encoder"],
                                scope=f"{self.scope}.encoder", store_intermediates=self.store_intermediates)
        self.seq_layers = [
            FxpSequenceLayer(modeldict=self.modeldict[f"layers_{idx}"], fxp_qconfig=self.fxp_qconfig["blocks"],
                             scope=f"{self.scope}.layers_{idx}", store_intermediates=self.store_intermediates,
                             mixer_cls=self.mixer_cls, layer_idx=idx, **{k: getattr(self, k) for k in _LAYER_KW})
            for idx in range(self.n_layers)]

    def forward(self, x, integration_timesteps: int = None):
        self.sow("intermediates", "pre_encoder", x)
        x = self.encoder(x)
        self.sow("intermediates", "encoder_output", x)
        x = fxp_relu(x)
        self.sow("intermediates", "encoder_output_relu", x)
        for idx, layer in enumerate(self.seq_layers):
            x = layer(x)
            self.sow("intermediates", f"layer_{idx}_output", x)
        return x

    def export(self):
        encoder_data = self.encoder.export()
        layers_data = [layer.export() for layer in self.seq_layers]
        data = {key: {f"layers_{idx}": layers_data[idx][key] for idx in range(self.n_layers)}
                for key in ["params", "intermediates", "qconfig"]}
        data["params"]["encoder"] = encoder_data["params"]
        data["intermediates"]["encoder"] = encoder_data["intermediates"]
        data["qconfig"]["encoder"] = encoder_data["qconfig"]
        data["intermediates"] = {**data["intermediates"], **self.last_intermediates()}
        return data


_STACK_KW = ("n_layers", "d_model", "dropout", "batchnorm", "prenorm", "bn_momentum", "glu_variant", "training",
             "step_rescale", "relufication", "fuse_batchnorm_linear", "q_config")


@dataclass
class FxpRegressionModel(FxpModule):  # :1380-1458
    mixer_cls: Any
    n_layers: int
    d_model: int
    batchnorm: bool = True
    prenorm: bool = False
    bn_momentum: float = 0.9
    glu_variant: str = "none"
    step_rescale: float = 1.0
    relufication: bool = False
    fuse_batchnorm_linear: bool = False
    q_config: Any = None
    dropout: float = 0.2
    training: bool = True
    d_output: int = None
    padded: bool = False
    # build-specific knobs (not in the reference): kernel selection and cross-rank exponent mode
    engine_flags: int = 0
    exponent_allreduce: Optional[Callable] = None

    def setup(self):
        assert self.batchnorm, "Only batchnorm is supported for now."
        assert self.dropout == 0.0, "Only dropout=0.0 is supported for now."
        assert not self.training, "Only training=False is supported for now."
        assert self.relufication, "Only relufication=True is supported for now."
        self.encoder = FxpStackedEncoderModel(
            modeldict=self.modeldict["encoder"], fxp_qconfig=self.fxp_qconfig, scope=self.scope + ".encoder",
            store_intermediates=self.store_intermediates, mixer_cls=self.mixer_cls,
            **{k: getattr(self, k) for k in _STACK_KW})
        self.decoder = FxpDense(modeldict=self.modeldict["decoder"], fxp_qconfig=self.fxp_qconfig["decoder"],
                                scope=self.scope + ".decoder", store_intermediates=self.store_intermediates)
        self._engine = None
        self._generic_engine = None

    # -- fused path -----------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(self.export(), flags=self.engine_flags)
        return self._engine

    def forward(self, x, integration_timesteps: int = 10):
        if self.padded:
            x, _ = x
        if self.store_intermediates:
            h = self.encoder(x, integration_timesteps)
            self.sow("intermediates", "encoder_output", h)
            y = self.decoder(h)
            self.sow("intermediates", "output", y)
            return y
        try:
            return self.engine().forward(x, allreduce=self.exponent_allreduce)
        except OverflowError:
            # the input holds values beyond its nominal bits (legal in the reference): 32-bit kernels
            from ._lib import MODEL_FORCE_GENERIC
            from .engine import Engine
            if self._generic_engine is None:
                self._generic_engine = Engine(self.export(), flags=self.engine_flags | MODEL_FORCE_GENERIC)
            return self._generic_engine.forward(x, allreduce=self.exponent_allreduce)

    def export(self):
        encoder_data, decoder_data = self.encoder.export(), self.decoder.export()
        return dict(
            params=dict(encoder=encoder_data["params"], decoder=decoder_data["params"]),
            qconfig=dict(encoder=encoder_data["qconfig"], decoder=decoder_data["qconfig"]),
            intermediates=dict(encoder=encoder_data["intermediates"], decoder=decoder_data["intermediates"],
                               **self.last_intermediates()))


def build_regression_model(modeldict, fxp_qconfig, n_layers: int, store_intermediates: bool = False,
                           glu_variant: str = "half1", **kw) -> FxpRegressionModel:
    """The model_cls(...) partial of fxprun.py:427-457 for the NDNS recipe."""
    H = modeldict["encoder"]["encoder"]["kernel"].shape[-1]
    P = modeldict["encoder"]["layers_0"]["mixer"]["Lambda_re"].shape[0]
    mixer_cls = FxpSSM.init_fn(H=H, P=P, discretization="zoh", conj_sym=True, q_config=QuantizationConfig.none(),
                               bidirectional=False, relufication=True, associative_scan=False)
    return FxpRegressionModel(
        modeldict=modeldict, fxp_qconfig=fxp_qconfig, scope="model", mixer_cls=mixer_cls, n_layers=n_layers,
        d_model=H, batchnorm=True, prenorm=True, bn_momentum=0.95, glu_variant=glu_variant, step_rescale=1.0,
        relufication=True, fuse_batchnorm_linear=False, q_config=QuantizationConfig.none(), dropout=0.0,
        training=False, d_output=modeldict["decoder"]["kernel"].shape[-1], padded=False,
        store_intermediates=store_intermediates, **kw)
