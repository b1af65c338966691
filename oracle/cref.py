"""ctypes binding of oracle/s5fxp_ref.c  --  TEST INFRASTRUCTURE ONLY (see that file's header).

Takes the INTEGER model in the reference's ``export()`` layout (``{"params": ..., "qconfig": ...}``,
sparseRNNs/fxpmodel.py:368-393,819-847,946-968,1163-1207,1441-1458) and runs the scalar C forward.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libs5fxp_ref.so")

I32P = C.POINTER(C.c_int32)


class RefDense(C.Structure):
    _fields_ = [("K", C.c_int32), ("M", C.c_int32), ("w", I32P), ("bias", I32P)] + [
        (n, C.c_int32) for n in ("w_exp", "b_bits", "b_exp", "inp_bits", "inp_exp", "out_bits", "out_exp")]


class RefSSM(C.Structure):
    _fields_ = [("H", C.c_int32), ("P", C.c_int32)] + [
        (n, I32P) for n in ("a_re", "a_im", "b_re", "b_im", "c_re", "c_im", "d")] + [
        (n, C.c_int32) for n in ("a_re_exp", "a_im_exp", "b_re_exp", "b_im_exp", "c_re_exp", "c_im_exp", "d_exp",
                                 "u_bits", "u_exp", "bu_re_bits", "bu_re_exp", "bu_im_bits", "bu_im_exp",
                                 "x_re_exp", "x_im_exp", "y_bits", "y_exp")]


class RefBN(C.Structure):
    _fields_ = [(n, I32P) for n in ("minus_mean", "invsq_var", "scale", "bias")] + [
        (n, C.c_int32) for n in ("mean_bits", "mean_exp", "isv_bits", "isv_exp", "scale_bits", "scale_exp",
                                 "bias_bits", "bias_exp")]


class RefLayer(C.Structure):
    _fields_ = [("bn", RefBN), ("ssm", RefSSM), ("out2", RefDense)] + [
        (n, C.c_int32) for n in ("l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp", "sig_x_exp",
                                 "sig_y_exp")] + [("lut", C.c_int32 * 8)]


class RefModel(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("enc", RefDense), ("layers", C.POINTER(RefLayer)), ("dec", RefDense)]


TRACE_KEYS = ("pre_s5", "u", "bu_re", "bu_im", "xs_re", "xs_im", "ys", "out2", "sigmoid", "post_glu", "residadd")


class RefLayerTrace(C.Structure):
    _fields_ = [(n, I32P) for n in TRACE_KEYS] + [("pre_s5_exp", C.c_int32), ("residadd_exp", C.c_int32)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "s5fxp_ref.c")):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.ref_forward.restype = C.c_int
        _lib.ref_forward.argtypes = [C.POINTER(RefModel), I32P, C.c_int, C.c_int, C.c_int, C.c_int, I32P,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(RefLayerTrace), C.c_int]
        _lib.ref_forward_state.restype = C.c_int
        _lib.ref_forward_state.argtypes = _lib.ref_forward.argtypes + [I32P]
        _lib.ref_num_threads.restype = C.c_int
    return _lib


def num_threads() -> int:
    return lib().ref_num_threads()


class CModel:
    """Owns contiguous int32 copies of every parameter array and the ctypes structs over them."""

    def __init__(self, export: dict, sigmoid_luts: Optional[List[np.ndarray]] = None):
        self._keep: list = []
        P, Q = export["params"], export["qconfig"]
        n_layers = len([k for k in P["encoder"] if k.startswith("layers_")])
        self.n_layers = n_layers
        self.layers = (RefLayer * n_layers)()
        for i in range(n_layers):
            lp, lq = P["encoder"][f"layers_{i}"], Q["encoder"][f"layers_{i}"]
            L = self.layers[i]
            L.out2 = self._dense(lp["out2"], lq["out2"])
            m, mq = lp["mixer"], lq["mixer"]
            s = L.ssm
            s.P, s.H = m["B_real"].shape
            for f, k in (("a_re", "A_real"), ("a_im", "A_imag"), ("b_re", "B_real"), ("b_im", "B_imag"),
                         ("c_re", "C_real"), ("c_im", "C_imag"), ("d", "D")):
                setattr(s, f, self._ptr(m[k]))
                setattr(s, f + "_exp", int(mq[f"{k}_exp"]))
            for f, k in (("u", "u"), ("bu_re", "Bu_re"), ("bu_im", "Bu_im"), ("y", "y")):
                setattr(s, f + "_bits", int(mq[f"{k}_bits"]))
                setattr(s, f + "_exp", int(mq[f"{k}_exp"]))
            s.x_re_exp, s.x_im_exp = int(mq["x_re_exp"]), int(mq["x_im_exp"])
            n, nq = lp["norm"], lq["norm"]
            bn = L.bn
            bn.minus_mean = self._ptr(-np.asarray(n["mean"], dtype=np.int64))
            bn.invsq_var = self._ptr(n["invsq_var"])
            bn.mean_bits, bn.mean_exp = int(nq["mean_bits"]), int(nq["mean_exp"])
            bn.isv_bits, bn.isv_exp = int(nq["invsq_var_bits"]), int(nq["invsq_var_exp"])
            if "scale" in n:
                bn.scale, bn.scale_bits, bn.scale_exp = self._ptr(n["scale"]), int(nq["scale_bits"]), int(nq["scale_exp"])
            if "bias" in n:
                bn.bias, bn.bias_bits, bn.bias_exp = self._ptr(n["bias"]), int(nq["bias_bits"]), int(nq["bias_exp"])
            for k in ("l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp"):
                setattr(L, k, int(lq["multgate"][k]))
            L.sig_x_exp, L.sig_y_exp = int(lq["sigmoid"]["x_exp"]), int(lq["sigmoid"]["y_exp"])
            if sigmoid_luts is not None:
                lut = sigmoid_luts[i]
            else:  # the LUT is a pure function of (x_exp, y_exp): fxpmodel.py:89-95
                from .fxp_oracle import sigmoid_lut
                lut = sigmoid_lut(L.sig_x_exp, L.sig_y_exp)
            for j in range(8):
                L.lut[j] = int(lut[j])
        self.model = RefModel()
        self.model.n_layers = n_layers
        self.model.enc = self._dense(P["encoder"]["encoder"], Q["encoder"]["encoder"])
        self.model.layers = C.cast(self.layers, C.POINTER(RefLayer))
        self.model.dec = self._dense(P["decoder"], Q["decoder"])
        self.H = self.model.enc.M
        self.P = self.layers[0].ssm.P if n_layers else 0
        self.d_out = self.model.dec.M

    def _ptr(self, a) -> I32P:
        arr = np.ascontiguousarray(np.asarray(a).astype(np.int64).astype(np.int32))
        self._keep.append(arr)
        return arr.ctypes.data_as(I32P)

    def _dense(self, p: dict, q: dict) -> RefDense:
        d = RefDense()
        d.K, d.M = p["weight"].shape
        d.w = self._ptr(p["weight"])
        d.bias = self._ptr(p["bias"]) if p.get("bias") is not None else None
        d.w_exp, d.b_bits, d.b_exp = int(q["weight_exp"]), int(q["bias_bits"]), int(q["bias_exp"])
        d.inp_bits, d.inp_exp = int(q["inp_bits"]), int(q["inp_exp"])
        d.out_bits, d.out_exp = int(q["out_bits"]), int(q["out_exp"])
        return d

    def forward(self, x: np.ndarray, x_bits: int, x_exp: int, trace: bool = False, nthreads: int = 0,
                state: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int, int, Optional[List[Dict[str, np.ndarray]]]]:
        """x: int32 (B,L,d_in) or (L,d_in).  Returns (y, y_bits, y_exp, traces).
        state: None, or a C-contiguous int32 array (n_layers, 2, B, P), read and replaced in place (the streaming carry)."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        shp = x.shape
        B, L = (1, shp[0]) if x.ndim == 2 else shp[:2]
        y = np.empty(shp[:-1] + (self.d_out,), dtype=np.int32)
        traces = None
        tr_structs = None
        if trace:
            tr_structs = (RefLayerTrace * self.n_layers)()
            traces = []
            for i in range(self.n_layers):
                d = {}
                for k in TRACE_KEYS:
                    w = self.P if k in ("bu_re", "bu_im", "xs_re", "xs_im") else self.H
                    d[k] = np.empty(shp[:-1] + (w,), dtype=np.int32)
                    setattr(tr_structs[i], k, d[k].ctypes.data_as(I32P))
                traces.append(d)
        yb, ye = C.c_int(0), C.c_int(0)
        if state is not None:
            assert state.dtype == np.int32 and state.flags["C_CONTIGUOUS"] and state.shape == (self.n_layers, 2, B, self.P)
        rc = lib().ref_forward_state(C.byref(self.model), x.ctypes.data_as(I32P), x_bits, x_exp, B, L,
                                     y.ctypes.data_as(I32P), C.byref(yb), C.byref(ye),
                                     C.cast(tr_structs, C.POINTER(RefLayerTrace)) if trace else None, nthreads,
                                     state.ctypes.data_as(I32P) if state is not None else None)
        if rc == -2:
            raise ValueError("negative / out-of-range shift (invalid result_exp)")
        if rc != 0:
            raise RuntimeError(f"ref_forward failed: {rc}")
        if trace:
            for i in range(self.n_layers):
                traces[i]["pre_s5_exp"] = tr_structs[i].pre_s5_exp
                traces[i]["residadd_exp"] = tr_structs[i].residadd_exp
        return y, yb.value, ye.value, traces
