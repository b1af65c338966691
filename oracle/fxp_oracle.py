"""CPU spec oracle for the fixed-point S5 inference path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the integer semantics of the reference's
``sparseRNNs/fxparray.py`` (FxpArray ops) and ``sparseRNNs/fxpmodel.py`` (the
non-fused-BatchNorm, ``use_lax_scan=True``, GLU "half1" forward).  It exists so
the HIP path can be checked; it is never the thing that is shipped or measured.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.

PARITY UNPINNED.  The reference has no tests, golden vectors or fixtures for this
path and JAX is not installed in the build container (``import jax`` raises
ModuleNotFoundError), so the reference itself cannot be run to pin this oracle.
It is pinned only by (a) known-answer values derived by hand from the reference's
formulas (tests/test_oracle_kat.py) and (b) bit-for-bit agreement with a second,
independently written scalar C restatement (oracle/s5fxp_ref.c).

JAX semantics encoded here (reference runs with jax_enable_x64 off):
  * every integer array is int32, + - * << wrap modulo 2**32, >> is arithmetic;
  * int32 -> float32 is round-to-nearest-even, x / 2**e is exact;
  * jnp.round is half-to-even; float32 -> int32 truncates (and saturates);
  * "compute_best" exponents come from float32 maxima followed by
    ceil(log2(.)) evaluated in float32.  We define log2 as the CORRECTLY ROUNDED
    float32 log2; XLA's own approximation may differ by 1 ulp, which can move the
    result for maxima within a few ulp above a power of two >= 16 (documented
    ambiguity; fixtures are checked to stay clear of it).

All citations are file:line into /root/reference/sparseRNNs/.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

I32 = np.int32
F32 = np.float32

FLOOR, CEIL, ROUND = 0, 1, 2  # fxparray.py:13-17


# --------------------------------------------------------------------------------------
# value type
# --------------------------------------------------------------------------------------
@dataclass
class Fx:
    """value = data / 2**exp ; saturation bounds from (bits, signed).  fxparray.py:33-38."""

    data: np.ndarray
    bits: int
    exp: int
    signed: bool = True

    def f32(self) -> np.ndarray:
        """fxparray.py:72-73  (int32 -> f32 RNE, exact power-of-two divide)."""
        return np.ldexp(self.data.astype(F32), -self.exp).astype(F32)

    def copy(self) -> "Fx":
        return Fx(self.data.copy(), self.bits, self.exp, self.signed)


def _i32(a) -> np.ndarray:
    return np.asarray(a).astype(I32)


def lo(bits: int, signed: bool = True) -> int:  # fxparray.py:329-330
    return -(1 << (bits - 1)) if signed else 0


def hi(bits: int, signed: bool = True) -> int:  # fxparray.py:333-334
    return (1 << (bits - 1)) - 1 if signed else (1 << bits) - 1


def sat(d: np.ndarray, bits: int, signed: bool = True) -> np.ndarray:
    """fxparray.py:346-357 (clip; the warning side effects are dropped)."""
    return np.clip(d, lo(bits, signed), hi(bits, signed)).astype(I32)


def shl(d: np.ndarray, s: int) -> np.ndarray:
    """int32 left shift with wrap."""
    assert 0 <= s < 32, f"left shift {s} is outside XLA's defined range"
    with np.errstate(over="ignore"):
        return (d.astype(np.int64) << s).astype(I32)  # astype wraps modulo 2**32


def asr(d: np.ndarray, s: int, mode: int = FLOOR) -> np.ndarray:
    """fxparray.py:274-284."""
    assert 0 <= s < 32, f"right shift {s} is outside XLA's defined range"
    d = d.astype(I32)
    with np.errstate(over="ignore"):
        if mode == FLOOR:
            return d >> I32(s)
        if mode == CEIL:
            return (d + I32((1 << s) - 1)) >> I32(s)
        if mode == ROUND:
            return (d + I32(1 << (s - 1))) >> I32(s)
    raise NotImplementedError


def mul32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """int32 * int32 -> low 32 bits."""
    return (a.astype(np.int64) * b.astype(np.int64)).astype(I32)


def add32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return (a.astype(np.int64) + b.astype(np.int64)).astype(I32)


def sub32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return (a.astype(np.int64) - b.astype(np.int64)).astype(I32)


def matmul32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """int32 @ int32 with int32 accumulation (wrap).  fxparray.py:662.

    Accumulating in int64 and truncating is identical modulo 2**32 as long as the
    int64 sum itself cannot overflow: |a|<2**31, |b|<2**31 would, so split b.
    """
    a64 = a.astype(np.int64)
    b64 = b.astype(np.int64)
    if np.abs(a64).max(initial=0) < (1 << 31) and np.abs(b64).max(initial=0) < (1 << 16):
        # |sum| <= K * 2**47 : safe for K < 2**16
        return (a64 @ b64).astype(I32)
    b_lo = b64 & 0xFFFF
    b_hi = b64 >> 16
    r = (a64 @ b_lo) + (((a64 @ b_hi) & 0xFFFFFFFF) << 16)
    return r.astype(I32)


def f32_to_i32(x: np.ndarray) -> np.ndarray:
    """XLA convert f32 -> s32: truncate toward zero, saturating."""
    x = np.asarray(x, dtype=F32)
    y = np.trunc(x.astype(np.float64))
    y = np.clip(y, -(2.0**31), 2.0**31 - 1)
    return y.astype(np.int64).astype(I32)


def ceil_log2_f32(v) -> int:
    """int(ceil(log2(v))) with log2 the correctly rounded float32 function (see header)."""
    v = F32(v)
    assert v > 0
    return int(np.ceil(F32(np.log2(np.float64(v)))))


def intbits_f32(m, eps) -> int:
    """max(0, int(ceil(log2(m + eps)))) in float32.  fxparray.py:421-425, 603-607."""
    return max(0, ceil_log2_f32(F32(F32(m) + F32(eps))))


# --------------------------------------------------------------------------------------
# fxparray.py ops
# --------------------------------------------------------------------------------------
# Test hook for the multi-rank protocol (SURVEY.md §8e mode A): when set, every compute_best op passes
# its float32 maxima through it (e.g. an all_reduce(MAX) over ranks) before choosing the exponent.
MAX_EXCHANGE = None


def _exchange_max(v: np.ndarray) -> np.ndarray:
    return v if MAX_EXCHANGE is None else np.asarray(MAX_EXCHANGE(v), dtype=F32)
def from_fp(x, bits=16, exp=8, signed=True, mode=FLOOR) -> Fx:
    """fxparray.py:287-307.  x is float32."""
    x = np.asarray(x, dtype=F32)
    xi = (x * F32(1 << exp)).astype(F32)
    if not signed and np.any(xi < 0):
        xi = np.abs(xi)
    if mode == ROUND:
        r = np.rint(xi)  # half to even, fxparray.py:24
    elif mode == CEIL:
        r = np.ceil(xi)
    elif mode == FLOOR:
        r = np.floor(xi)
    else:
        raise NotImplementedError
    return Fx(sat(f32_to_i32(r), bits, signed), bits, exp, signed)


def change_exp(a: Fx, new_exp: int, mode: int = FLOOR) -> Fx:
    """fxparray.py:310-326.  NB: no clip when the exponent is unchanged; otherwise clip at
    the operand's CURRENT bits."""
    if new_exp == a.exp:
        return a.copy()
    if new_exp > a.exp:
        d = shl(a.data, new_exp - a.exp)
    else:
        d = asr(a.data, a.exp - new_exp, mode)
    return Fx(sat(d, a.bits, a.signed), a.bits, new_exp, a.signed)


def change_cfg(a: Fx, new_bits: int, new_exp: int, new_signed: bool = True, mode: int = FLOOR) -> Fx:
    """fxparray.py:232-271."""
    if a.bits == new_bits and a.exp == new_exp and a.signed == new_signed:
        return a
    r = change_exp(a, new_exp, mode)
    if r.bits > new_bits:
        r = Fx(sat(r.data, new_bits, r.signed), new_bits, r.exp, r.signed)
    else:
        r.bits = new_bits
    if (not r.signed) and new_signed:
        r = Fx(sat(r.data, r.bits, True), r.bits, r.exp, True)
    else:
        r.signed = new_signed
    return r


def add(a: Fx, b: Fx, result_bits: Optional[int] = None, result_exp=None, mode: int = FLOOR) -> Fx:
    """fxparray.py:386-466.  result_exp: int or "compute_best".  The result_exp=None branch
    with unequal exponents (fxparray.py:414-419) is never reached by the model and is not
    restated (its operator precedence makes it a different function)."""
    signed = a.signed or b.signed
    if result_bits is None:
        result_bits = max(a.bits, b.bits)
    if result_exp is None:
        assert a.exp == b.exp, "unequal-exponent default add is not part of the hot path"
        result_exp = a.exp
        d = add32(a.data, b.data)
    elif isinstance(result_exp, str):
        assert result_exp == "compute_best"
        fa, fb = a.f32(), b.f32()
        mx3 = _exchange_max(np.array([np.abs((fa + fb).astype(F32)).max(), np.abs(fa).max(), np.abs(fb).max()], dtype=F32))
        m = mx3[0]
        ib = intbits_f32(m, 1e-6)
        result_exp = result_bits - ib - (1 if signed else 0)
        ia = max(intbits_f32(mx3[1], 1e-8), intbits_f32(mx3[2], 1e-8))
        agg_exp = max(a.exp, b.exp)
        agg_bits = ia + agg_exp + (1 if signed else 0)
        ac = change_cfg(a, max(agg_bits, a.bits), agg_exp, signed)
        bc = change_cfg(b, max(agg_bits, b.bits), agg_exp, signed)
        d = add32(ac.data, bc.data)
        de = result_exp - agg_exp
        if de > 0:
            d = shl(d, de)
        elif de < 0:
            d = asr(d, -de)
    else:
        d = add32(change_exp(a, result_exp, mode).data, change_exp(b, result_exp, mode).data)
    return Fx(sat(d, result_bits, signed), result_bits, int(result_exp), signed)


def neg(a: Fx) -> Fx:
    """-1 * data, unclipped.  fxparray.py:374."""
    return Fx(mul32(a.data, np.asarray(-1, dtype=I32)), a.bits, a.exp, a.signed)


def sub(a: Fx, b: Fx, result_bits=None, result_exp=None, mode=FLOOR) -> Fx:
    """fxparray.py:360-383."""
    return add(a, neg(b), result_bits, result_exp, mode)


def mul(a: Fx, b: Fx, result_bits: Optional[int] = None, result_exp=None, mode: int = FLOOR) -> Fx:
    """fxparray.py:573-637.  The ">30 bit -> int64" branch (611-616) is a no-op under default
    JAX (int64 requests become int32)."""
    signed = a.signed or b.signed
    if result_bits is None:
        result_bits = max(a.bits, b.bits)
    if result_exp is None:
        result_exp = max(a.exp, b.exp)
    elif isinstance(result_exp, str):
        assert result_exp == "compute_best"
        m = _exchange_max(np.array([np.abs((a.f32() * b.f32()).astype(F32)).max()], dtype=F32))[0]
        ib = intbits_f32(m, 1e-6)
        result_exp = result_bits - ib - (1 if signed else 0)
    rshift = a.exp + b.exp - result_exp
    if rshift < 0:
        raise ValueError(f"invalid result_exp: {result_exp}")
    d = asr(mul32(a.data, b.data), rshift, mode)
    return Fx(sat(d, result_bits, signed), result_bits, int(result_exp), signed)


def matmul(a: Fx, b: Fx, result_bits: Optional[int] = None, result_exp: Optional[int] = None, mode=FLOOR) -> Fx:
    """fxparray.py:640-678."""
    signed = a.signed or b.signed
    if result_bits is None:
        result_bits = max(a.bits, b.bits)
    if result_exp is None:
        result_exp = max(a.exp, b.exp)
    raw = matmul32(a.data, b.data)
    rshift = a.exp + b.exp - result_exp
    if rshift < 0:
        # the reference passes a negative shift to XLA (undefined); the build treats it as an error
        raise ValueError(f"negative matmul shift {rshift}")
    return Fx(sat(asr(raw, rshift, mode), result_bits, signed), result_bits, result_exp, signed)


# --------------------------------------------------------------------------------------
# fxpmodel.py pieces
# --------------------------------------------------------------------------------------
def relu(a: Fx) -> Fx:
    """fxpmodel.py:53-63."""
    return Fx(np.maximum(a.data, I32(0)), a.bits, a.exp, a.signed)


def complex_relu(re: Fx, im: Fx) -> Tuple[Fx, Fx]:
    """fxpmodel.py:30-45: jax.nn.relu of a complex64 array = lexicographic maximum(z, 0);
    values round-trip through float32."""
    fr = re.data.astype(F32)
    fi = im.data.astype(F32)
    keep = (fr > 0) | ((fr == 0) & (fi > 0))
    r = np.where(keep, f32_to_i32(fr), I32(0)).astype(I32)
    i = np.where(keep, f32_to_i32(fi), I32(0)).astype(I32)
    return Fx(r, re.bits, re.exp, re.signed), Fx(i, im.bits, im.exp, im.signed)


def sigmoid_lut(x_exp: int, y_exp: int, x_extra: int = 3, n_exp: int = 3) -> np.ndarray:
    """fxpmodel.py:89-95 (``1 << a + b`` parses as ``1 << (a + b)``)."""
    x = np.linspace(0, 1 << (x_exp + x_extra), (1 << n_exp) + 1, dtype=F32)[:-1].astype(I32)
    xf = (x.astype(F32) / F32(1 << x_exp)).astype(F32)
    s = (F32(1) / (F32(1) + np.exp(-xf).astype(F32))).astype(F32)
    return f32_to_i32(np.rint((s * F32(1 << y_exp)).astype(F32)) - F32(1 << (y_exp - 1)))


def sigmoid_apply(x: Fx, x_exp: int, y_exp: int, lut: np.ndarray, n_exp: int = 3) -> Fx:
    """fxpmodel.py:97-144."""
    xx = change_exp(x, x_exp).data
    sign = np.where(xx > 0, I32(1), I32(-1))
    a = np.abs(xx).astype(I32)
    delta = I32(1 << x_exp)
    ind = np.minimum(a >> I32(x_exp), I32((1 << n_exp) - 2))
    mu = a & I32((1 << x_exp) - 1)
    half = add32(asr(mul32(delta - mu, lut[ind]), x_exp), asr(mul32(mu, lut[ind + 1]), x_exp))
    yy = add32(np.asarray(1 << (y_exp - 1), dtype=I32), mul32(sign, half))
    return Fx(yy, x.bits, y_exp, True)


def scan(bu_re: Fx, bu_im: Fx, a_re: Fx, a_im: Fx, x_re_exp: int, x_im_exp: int, x0=None) -> Tuple[np.ndarray, np.ndarray]:
    """fxpmodel.py:147-208.  bu_*: (..., L, P); a_*: (P,).  Sequential in L, no clip.
    x0: optional (re, im) int32 arrays (..., P), the carry the step function starts from (fxpmodel.py:147-172 makes the
    carry explicit; recurrent_loop always passes zeros, :196-207) -- the streaming API's state."""
    L = bu_re.data.shape[-2]

    def shiftto(v, e_from, e_to):  # fxpmodel.py:158-167
        return asr(v, e_from - e_to) if e_from > e_to else shl(v, e_to - e_from)

    br = shiftto(bu_re.data, bu_re.exp, x_re_exp)
    bi = shiftto(bu_im.data, bu_im.exp, x_im_exp)
    xr = np.zeros(bu_re.data.shape[:-2] + bu_re.data.shape[-1:], dtype=I32)
    xi = np.zeros_like(xr)
    if x0 is not None:
        xr, xi = _i32(x0[0]).reshape(xr.shape).copy(), _i32(x0[1]).reshape(xi.shape).copy()
    out_r = np.empty_like(bu_re.data)
    out_i = np.empty_like(bu_im.data)
    ar, ai = a_re.data, a_im.data
    for t in range(L):
        nr = add32(sub32(asr(mul32(ar, xr), a_re.exp), asr(mul32(ai, xi), a_re.exp)), br[..., t, :])
        ni = add32(add32(asr(mul32(ar, xi), a_im.exp), asr(mul32(ai, xr), a_im.exp)), bi[..., t, :])
        xr, xi = nr, ni
        out_r[..., t, :] = xr
        out_i[..., t, :] = xi
    return out_r, out_i


def discretize_zoh(Lambda: np.ndarray, B_tilde: np.ndarray, Delta: np.ndarray):
    """model/ssm.py:37-50 in complex64 / float32."""
    Lambda = Lambda.astype(np.complex64)
    Delta = Delta.astype(F32)
    Lambda_bar = np.exp((Lambda * Delta).astype(np.complex64)).astype(np.complex64)
    ident = np.ones(Lambda.shape[0], dtype=F32)
    B_bar = ((np.complex64(1) / Lambda * (Lambda_bar - ident)).astype(np.complex64)[..., None] * B_tilde.astype(np.complex64))
    return Lambda_bar, B_bar.astype(np.complex64)


# --------------------------------------------------------------------------------------
# Model: setup (float -> int) and forward.  Names of intermediates follow SURVEY App. A.
# --------------------------------------------------------------------------------------
class Dense:
    """fxpmodel.py:291-366."""

    def __init__(self, md: dict, qc: dict):
        self.qc = qc
        self.weight = from_fp(md["kernel"], qc["w_bits"], qc["w_exp"], True, ROUND)
        self.bias = from_fp(md["bias"], qc["b_bits"], qc["b_exp"], True, ROUND) if md.get("bias") is not None else None

    def __call__(self, x: Fx) -> Fx:
        qc = self.qc
        if x.bits > qc["inp_bits"] or x.exp > qc["inp_exp"]:
            x = change_cfg(x, qc["inp_bits"], qc["inp_exp"], True)
        y = matmul(x, self.weight, qc["out_bits"], qc["out_exp"])
        if self.bias is not None:
            y = add(y, self.bias, qc["out_bits"], qc["out_exp"])
        return y


class SSM:
    """fxpmodel.py:396-794, non-fused BN branch only."""

    def __init__(self, md: dict, qc: dict, step_rescale: float = 1.0):
        self.qc = qc
        B_tilde = (md["B"][..., 0] + 1j * md["B"][..., 1]).astype(np.complex64)
        Lam = (md["Lambda_re"] + 1j * md["Lambda_im"]).astype(np.complex64)
        step = (F32(step_rescale) * np.exp(md["log_step"][:, 0].astype(F32))).astype(F32)
        Lbar, Bbar = discretize_zoh(Lam, B_tilde, step)
        C = (md["C"][..., 0] + 1j * md["C"][..., 1]).astype(np.complex64)
        w = qc["weights"]
        q = lambda v, k: from_fp(np.asarray(v, dtype=F32), w[k]["bits"], w[k]["exp"], True, ROUND)
        self.A_re, self.A_im = q(Lbar.real, "A_re"), q(Lbar.imag, "A_im")
        self.B_re, self.B_im = q(Bbar.real, "B_re"), q(Bbar.imag, "B_im")
        self.C_re, self.C_im = q(C.real, "C_re"), q(C.imag, "C_im")
        self.D = q(md["D"], "D")

    def __call__(self, x: Fx, inter: Optional[dict] = None, state=None):
        """state: None, or a two-element list [re, im] of (..., P) int32 arrays: the recurrence starts from it and it is
        replaced by the state after the last step (raw, before the complex ReLU)."""
        act = self.qc["activations"]
        u = change_cfg(x, act["u"]["bits"], act["u"]["exp"], True)
        tr = lambda f: Fx(f.data.T, f.bits, f.exp, f.signed)
        bu_re = matmul(u, tr(self.B_re), act["Bu_re"]["bits"], act["Bu_re"]["exp"])
        bu_im = matmul(u, tr(self.B_im), act["Bu_im"]["bits"], act["Bu_im"]["exp"])
        xr, xi = scan(bu_re, bu_im, self.A_re, self.A_im, act["x_re"]["exp"], act["x_im"]["exp"], x0=state)
        if state is not None:
            state[0], state[1] = xr[..., -1, :].copy(), xi[..., -1, :].copy()
        xs_re = Fx(xr, act["x_re"]["bits"], act["x_re"]["exp"], True)
        xs_im = Fx(xi, act["x_im"]["bits"], act["x_im"]["exp"], True)
        rr, ri = complex_relu(xs_re, xs_im)
        yb, ye = act["y"]["bits"], act["y"]["exp"]
        cx = sub(matmul(rr, tr(self.C_re), yb, ye), matmul(ri, tr(self.C_im), yb, ye), yb, ye)
        cx2 = Fx(mul32(cx.data, np.asarray(2, dtype=I32)), cx.bits, cx.exp, cx.signed)  # fxpmodel.py:765-767
        du = mul(self.D, u, yb, ye)
        ys = add(cx2, du, yb, ye)
        if inter is not None:
            inter.update(u=u, Bu_re=bu_re, Bu_im=bu_im, xs_re=xs_re, xs_im=xs_im, xs_relu_re=rr, xs_relu_im=ri,
                         Cxs=cx, Cxs2=cx2, Du=du, ys=ys)
        return ys, (rr, ri)


class BatchNorm:
    """fxpmodel.py:850-944."""

    def __init__(self, md: dict, qc: dict, bn_eps: float = 1e-5):
        q = lambda v, k: from_fp(np.asarray(v, dtype=F32), qc[k]["bits"], qc[k]["exp"], True, ROUND)
        self.minus_mean = q(F32(-1) * md["mean"].astype(F32), "mean")
        self.invsq_var = q(F32(1.0) / np.sqrt(md["var"].astype(F32) + F32(bn_eps)), "invsq_var")
        self.bias = q(md["bias"], "bias") if "bias" in md else None
        self.scale = q(md["scale"], "scale") if "scale" in md else None

    def __call__(self, x: Fx, inter: Optional[dict] = None) -> Fx:
        t1 = add(x, self.minus_mean, result_exp="compute_best")
        t2 = mul(t1, self.invsq_var, result_exp="compute_best")
        t = t2
        t3 = t4 = None
        if self.scale is not None:
            t = t3 = mul(t, self.scale, result_exp="compute_best")
        if self.bias is not None:
            t = t4 = add(t, self.bias, result_exp="compute_best")
        if inter is not None:
            inter.update(norm_input_minus_mean=t1, norm_output_raw=t2, norm_output=t)
            if t3 is not None:
                inter["norm_output_scaled"] = t3
            if t4 is not None:
                inter["norm_output_scaled_bias"] = t4
        return t


class SequenceLayer:
    """fxpmodel.py:971-1161, prenorm + batchnorm + relufication, glu_variant "half1"."""

    def __init__(self, md: dict, qc_blocks: dict, layer_idx: int):
        keys = [k for k in qc_blocks if k.startswith("layers_")]
        qc = qc_blocks[f"layers_{layer_idx}"] if keys else qc_blocks  # fxpmodel.py:988-993
        self.qc = qc
        self.norm = BatchNorm(md["norm"], qc["norm"])
        self.mixer = SSM(md["seq"] if "seq" in md else md["mixer"], qc["ssm"])
        self.out2 = Dense(md["out2"], qc["out2"])
        self.sig_x_exp = min(qc["out2"]["out_exp"], 6)  # fxpmodel.py:1097-1103
        self.sig_y_exp = qc["out2"]["out_bits"] - 2
        self.lut = sigmoid_lut(self.sig_x_exp, self.sig_y_exp)

    def __call__(self, x: Fx, inter: Optional[dict] = None, state=None) -> Fx:
        mg = self.qc["multgate"]
        skip = x
        n_i = {} if inter is not None else None
        m_i = {} if inter is not None else None
        t = self.norm(x, n_i)
        y, _ = self.mixer(t, m_i, state)
        x1 = relu(y)
        g_in = self.out2(x1)
        g = sigmoid_apply(g_in, self.sig_x_exp, self.sig_y_exp, self.lut)
        z = mul(change_cfg(x1, mg["l_bits"], mg["l_exp"], True), change_cfg(g, mg["r_bits"], mg["r_exp"], True),
                mg["res_bits"], mg["res_exp"])
        r = add(z, skip, mg["res_bits"], "compute_best")
        out = relu(r)
        if inter is not None:
            inter.update(ssm_input=skip, pre_s5=t, norm=n_i, mixer=m_i, pre_GLU=y, out2=g_in, out2_sigmoid=g,
                         post_GLU=z, residadd=r, output=out)
        return out


class RegressionModel:
    """fxpmodel.py:1210-1271, 1380-1439."""

    def __init__(self, modeldict: dict, fxp_qconfig: dict, n_layers: int):
        enc = modeldict["encoder"]
        self.encoder = Dense(enc["encoder"], fxp_qconfig["encoder"])
        self.layers = [SequenceLayer(enc[f"layers_{i}"], fxp_qconfig["blocks"], i) for i in range(n_layers)]
        self.decoder = Dense(modeldict["decoder"], fxp_qconfig["decoder"])

    def __call__(self, x: Fx, inter: Optional[dict] = None, state=None) -> Fx:
        """state: None, or a list with one [re, im] pair per layer (see SSM.__call__), updated in place: feeding a
        sequence chunk by chunk with the same list is the streaming use (every chunk is its own compute_best batch)."""
        e = self.encoder(x)
        h = relu(e)
        if inter is not None:
            inter["pre_encoder"], inter["encoder_output"], inter["encoder_output_relu"] = x, e, h
        for i, layer in enumerate(self.layers):
            li = {} if inter is not None else None
            h = layer(h, li, state[i] if state is not None else None)
            if inter is not None:
                inter[f"layers_{i}"] = li
        y = self.decoder(h)
        if inter is not None:
            inter["output"] = y
        return y

    def zero_state(self, batch_shape=()) -> list:
        P = self.layers[0].mixer.A_re.data.shape[0]
        return [[np.zeros(tuple(batch_shape) + (P,), dtype=I32), np.zeros(tuple(batch_shape) + (P,), dtype=I32)] for _ in self.layers]

    # -- integer export: the layout of fxpmodel.py export() (368-393, 819-847, 946-968, 1163-1207)
    def export(self) -> dict:
        def dense(d: Dense):
            return dict(params=dict(weight=d.weight.data, bias=d.bias.data),
                        qconfig=dict(weight_exp=d.weight.exp, weight_bits=d.weight.bits, bias_exp=d.bias.exp,
                                     bias_bits=d.bias.bits, inp_bits=d.qc["inp_bits"], inp_exp=d.qc["inp_exp"],
                                     out_bits=d.qc["out_bits"], out_exp=d.qc["out_exp"]))

        out = dict(params=dict(encoder=OrderedDict(), decoder=None), qconfig=dict(encoder=OrderedDict(), decoder=None))
        e = dense(self.encoder)
        out["params"]["encoder"]["encoder"], out["qconfig"]["encoder"]["encoder"] = e["params"], e["qconfig"]
        for i, l in enumerate(self.layers):
            m = l.mixer
            mp, mq = {}, {}
            for k, v in dict(A_real=m.A_re, A_imag=m.A_im, B_real=m.B_re, B_imag=m.B_im, C_real=m.C_re, C_imag=m.C_im,
                             D=m.D).items():
                mp[k] = v.data
                mq[f"{k}_bits"], mq[f"{k}_exp"] = v.bits, v.exp
            for k in ["u", "Bu_re", "Bu_im", "x_re", "x_im", "y"]:
                mq[f"{k}_bits"], mq[f"{k}_exp"] = m.qc["activations"][k]["bits"], m.qc["activations"][k]["exp"]
            n = l.norm
            np_ = dict(mean=mul32(n.minus_mean.data, np.asarray(-1, dtype=I32)), invsq_var=n.invsq_var.data)
            nq = dict(mean_bits=n.minus_mean.bits, mean_exp=n.minus_mean.exp, invsq_var_bits=n.invsq_var.bits,
                      invsq_var_exp=n.invsq_var.exp)
            if n.bias is not None:
                np_["bias"], nq["bias_bits"], nq["bias_exp"] = n.bias.data, n.bias.bits, n.bias.exp
            if n.scale is not None:
                np_["scale"], nq["scale_bits"], nq["scale_exp"] = n.scale.data, n.scale.bits, n.scale.exp
            o2 = dense(l.out2)
            mg = {k: l.qc["multgate"][k] for k in ["l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp"]}
            out["params"]["encoder"][f"layers_{i}"] = dict(mixer=mp, out2=o2["params"], norm=np_)
            out["qconfig"]["encoder"][f"layers_{i}"] = dict(
                mixer=mq, out2=o2["qconfig"], norm=nq, multgate=mg,
                sigmoid=dict(x_exp=l.sig_x_exp, y_exp=l.sig_y_exp, x_extra=3, n_exp=3))
        d = dense(self.decoder)
        out["params"]["decoder"], out["qconfig"]["decoder"] = d["params"], d["qconfig"]
        return out


def flatten_intermediates(inter: dict, prefix: str = "") -> Dict[str, Fx]:
    """{'layers_0': {'mixer': {'ys': Fx}}} -> {'layers_0.mixer.ys': Fx}."""
    flat: Dict[str, Fx] = {}
    for k, v in inter.items():
        if isinstance(v, dict):
            flat.update(flatten_intermediates(v, f"{prefix}{k}."))
        elif v is not None:
            flat[f"{prefix}{k}"] = v
    return flat
