"""ctypes binding of libs5fxp.so (the C ABI in include/s5fxp.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing this module raises, and every op in the package
fails with it.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64; libs5fxp.so links the same SONAME.  Whichever is loaded first is the HIP
# runtime of the process, and the streams / device pointers handed to the C ABI are torch's: torch must come first,
# or the library binds /opt/rocm's runtime and every launch fails with an invalid-handle error.
import torch  # noqa: F401  (load order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# S5FXP_LIB: developer override to load an experimental build of the same library (tools/build_variant.sh)
LIB_PATH = os.environ.get("S5FXP_LIB") or os.path.join(_HERE, "libs5fxp.so")

I32P = C.POINTER(C.c_int32)
VOIDP = C.c_void_p

S5FXP_OK, S5FXP_EBADARG, S5FXP_ENEGSHIFT, S5FXP_EUNSUPPORTED, S5FXP_EHIP, S5FXP_EWORKSPACE = 0, -1, -2, -3, -4, -5
ST_NEGSHIFT, ST_NEGEXP, ST_WIDE_STATE, ST_WIDE_INPUT, ST_REDO = 1, 2, 4, 8, 16
FWD_DEFER_REDO, FWD_EXACT, FWD_NO_PAIR = 1, 2, 4
STATUS_WORDS = 128
PATH_GENERIC, PATH_FUSED = 1, 2   # status[2]
MODEL_DEFAULT, MODEL_FORCE_DENSE, MODEL_FORCE_CSR, MODEL_FORCE_GENERIC = 0, 1, 2, 4


class DenseDesc(C.Structure):
    _fields_ = [("K", C.c_int32), ("M", C.c_int32), ("weight", I32P), ("bias", I32P)] + [
        (n, C.c_int32) for n in ("w_bits", "w_exp", "b_bits", "b_exp", "inp_bits", "inp_exp", "out_bits", "out_exp")]


class SSMDesc(C.Structure):
    _fields_ = [("H", C.c_int32), ("P", C.c_int32)] + [
        (n, I32P) for n in ("A_re", "A_im", "B_re", "B_im", "C_re", "C_im", "D")] + [
        (n, C.c_int32) for n in (
            "A_re_bits", "A_re_exp", "A_im_bits", "A_im_exp", "B_re_bits", "B_re_exp", "B_im_bits", "B_im_exp",
            "C_re_bits", "C_re_exp", "C_im_bits", "C_im_exp", "D_bits", "D_exp",
            "u_bits", "u_exp", "Bu_re_bits", "Bu_re_exp", "Bu_im_bits", "Bu_im_exp",
            "x_re_bits", "x_re_exp", "x_im_bits", "x_im_exp", "y_bits", "y_exp")]


class NormDesc(C.Structure):
    _fields_ = [(n, I32P) for n in ("minus_mean", "invsq_var", "scale", "bias")] + [
        (n, C.c_int32) for n in ("mean_bits", "mean_exp", "invsq_var_bits", "invsq_var_exp", "scale_bits",
                                 "scale_exp", "bias_bits", "bias_exp")]


class LayerDesc(C.Structure):
    _fields_ = [("norm", NormDesc), ("ssm", SSMDesc), ("out2", DenseDesc)] + [
        (n, C.c_int32) for n in ("l_bits", "l_exp", "r_bits", "r_exp", "res_bits", "res_exp", "sig_x_exp",
                                 "sig_y_exp")] + [("lut", C.c_int32 * 8)]


class ModelDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("encoder", DenseDesc), ("layers", C.POINTER(LayerDesc)),
                ("decoder", DenseDesc)]


TRACE_FIELDS = ("pre_s5", "u", "Bu_re", "Bu_im", "xs_re", "xs_im", "ys", "out2", "out2_sigmoid", "post_GLU",
                "residadd")


class LayerTrace(C.Structure):
    _fields_ = [(n, VOIDP) for n in TRACE_FIELDS]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, VOIDP, VOIDP, C.c_int, VOIDP)


class ForwardOpts(C.Structure):
    _fields_ = [("allreduce", ALLREDUCE_FN), ("allreduce_ctx", VOIDP), ("scan_events", C.POINTER(VOIDP)),
                ("flags", C.c_int32), ("state_in", VOIDP), ("state_out", VOIDP), ("groups", C.c_int32),
                ("gate_events", C.POINTER(VOIDP))]


class S5FxpError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built.  Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    i, i64, p = C.c_int, C.c_int64, VOIDP
    sig = {
        "s5fxp_version": (i, []),
        "s5fxp_strerror": (C.c_char_p, [i]),
        "s5fxp_from_fp": (i, [p, p, i64, i, i, i, p]),
        "s5fxp_to_float": (i, [p, p, i64, i, p]),
        "s5fxp_change_cfg": (i, [p, p, i64, i, i, i, i, p]),
        "s5fxp_dense": (i, [p, p, p, p, i64, i, i, i, i, i, i, i, i, i, p]),
        "s5fxp_dense_csr": (i, [p, p, p, p, p, p, i64, i, i, i, i, i, i, i, i, i, p]),
        "s5fxp_add": (i, [p, p, p, i64, i64, i, i, i, i, i, i, i, p]),
        "s5fxp_mul": (i, [p, p, p, i64, i64, i, i, i, i, p]),
        "s5fxp_add_cb": (i, [p, p, p, i64, i64, i, i, i, i, i, p, p, p]),
        "s5fxp_mul_cb": (i, [p, p, p, i64, i64, i, i, i, p, p, p]),
        "s5fxp_relu": (i, [p, p, p, p, i64, p]),
        "s5fxp_sigmoid": (i, [p, p, i64, i, i, i, i, I32P, p]),
        "s5fxp_scan": (i, [p, p, p, p, p, p, i, i, i, i, i, i, i, i, i, i, p]),
        "s5fxp_assoc_scan_c64": (i, [p, p, p, p, p, i, i, i, i, p]),
        "s5fxp_model_blob_bytes": (C.c_size_t, [C.POINTER(ModelDesc)]),
        "s5fxp_model_create": (i, [C.POINTER(ModelDesc), p, C.c_size_t, i, p, C.POINTER(p)]),
        "s5fxp_model_destroy": (None, [p]),
        "s5fxp_workspace_bytes": (C.c_size_t, [p, i, i]),
        "s5fxp_model_forward": (i, [p, p, i, i, i, i, p, p, C.c_size_t, p, C.POINTER(LayerTrace), C.POINTER(ForwardOpts), p]),
        "s5fxp_layer_forward": (i, [p, i, p, i, i, i, i, p, p, p, C.c_size_t, p, C.POINTER(LayerTrace), C.POINTER(ForwardOpts), p]),
        "s5fxp_model_layer_out_bits": (i, [p, i]),
        "s5fxp_model_live_states": (i, [p, i]),
        "s5fxp_model_out_exp": (i, [p]),
        "s5fxp_model_out_bits": (i, [p]),
        "s5fxp_model_is_fast": (i, [p]),
        "s5fxp_model_recurrence_kernel": (i, [p, i]),
        "s5fxp_model_recurrence_xmax": (i, [p, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale: rebuild
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
EXPORTED_SYMBOLS = ("s5fxp_version s5fxp_strerror s5fxp_from_fp s5fxp_to_float s5fxp_change_cfg s5fxp_dense s5fxp_dense_csr s5fxp_add "
                    "s5fxp_mul s5fxp_add_cb s5fxp_mul_cb s5fxp_relu s5fxp_sigmoid s5fxp_scan s5fxp_assoc_scan_c64 s5fxp_model_blob_bytes "
                    "s5fxp_model_create s5fxp_model_destroy s5fxp_workspace_bytes s5fxp_model_forward s5fxp_layer_forward s5fxp_model_layer_out_bits s5fxp_model_live_states "
                    "s5fxp_model_out_exp s5fxp_model_out_bits s5fxp_model_is_fast s5fxp_model_recurrence_kernel s5fxp_model_recurrence_xmax").split()


def check(rc: int, what: str = "") -> None:
    """Maps C-ABI error codes onto the exceptions the reference raises for the same condition."""
    if rc == S5FXP_OK:
        return
    msg = f"{what}: {lib.s5fxp_strerror(rc).decode()}" if what else lib.s5fxp_strerror(rc).decode()
    if rc == S5FXP_ENEGSHIFT:
        raise ValueError(msg)  # fxparray.py:619-621
    if rc == S5FXP_EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise S5FxpError(msg)
