"""FxpArray and its ops on MI355X: the host-side mirror of the reference's ``sparseRNNs/fxparray.py``.

Same names, argument meaning and error behaviour as the reference; ``data`` is a ``torch.int32``
ROCm tensor and every op is one or more launches of the HIP kernels behind the C ABI in
``include/s5fxp.h`` (PyTorch only provides device memory and streams).  There is no CPU path:
a tensor that is not on a GPU is moved to the current one, and without a GPU every op raises.

Citations are file:line into /root/reference/sparseRNNs/fxparray.py.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Optional, Union

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


class RoundingMode(Enum):  # :13-17
    FLOOR = 0
    CEIL = 1
    ROUND = 2
    STOCHASTIC = 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(x, dtype) -> torch.Tensor:
    """To a contiguous tensor of `dtype` on the current GPU (fails loudly without one)."""
    if not torch.cuda.is_available():
        raise RuntimeError("sparsernns_amd needs a ROCm GPU: there is no CPU implementation of the fxp ops")
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    if not x.is_cuda:
        x = x.cuda()
    if x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


@dataclass
class FxpArray:
    """value = data / 2**exp; (bits, signed) give the saturation bounds.  :33-167"""

    data: Optional[torch.Tensor] = None
    bits: int = 16
    exp: int = 8
    signed: bool = True

    def __post_init__(self):
        if self.data is not None and not (isinstance(self.data, torch.Tensor) and self.data.is_cuda
                                          and self.data.dtype == torch.int32):
            self.data = _dev(self.data, torch.int32)

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return self.data.ndim

    @property
    def dtype(self):
        return self.data.dtype

    def copy(self) -> "FxpArray":
        return FxpArray(self.data.clone(), self.bits, self.exp, self.signed)

    def minval(self) -> int:
        return fxp_minval(self)

    def maxval(self) -> int:
        return fxp_maxval(self)

    def clip(self, do_warn: bool = False, warn_prefix: str = ""):
        return fxp_clip(self, do_warn=do_warn, warn_prefix=warn_prefix)

    def is_valid(self, do_warn: bool = False) -> bool:
        return fxp_isvalid(self, do_warn=do_warn)

    def to_float(self) -> torch.Tensor:  # :72-73
        x = self.data.contiguous()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        check(lib.s5fxp_to_float(_ptr(x), _ptr(y), x.numel(), self.exp, _stream()), "to_float")
        return y

    def numpy(self) -> np.ndarray:
        return self.data.cpu().numpy()

    def change_exp(self, new_exp: int, round_mode: RoundingMode = RoundingMode.FLOOR, warn_on_clip: bool = True):
        return fxp_change_exp(self, new_exp=new_exp, round_mode=round_mode, warn_on_clip=warn_on_clip)

    def change_cfg(self, new_bits: int, new_exp: int, new_signed: bool,
                   round_mode: RoundingMode = RoundingMode.FLOOR, warn_on_clip: bool = True):
        return fxp_change_cfg(self, new_bits, new_exp, new_signed, round_mode, warn_on_clip)

    def __eq__(self, val):  # :100-109
        if not isinstance(val, FxpArray):
            raise TypeError(f"unsupported type for comparison with FxpArray: '{type(val)}'")
        return (self.bits == val.bits and self.exp == val.exp and self.signed == val.signed
                and self.data.shape == val.data.shape and bool(torch.equal(self.data, val.data)))

    def __add__(self, other):
        return fxp_add(self, other)

    def __sub__(self, other):
        return fxp_sub(self, other)

    def __mul__(self, other):
        return fxp_mul(self, other)

    def __matmul__(self, other):
        return fxp_matmul(self, other)

    def __getitem__(self, key) -> "FxpArray":
        return FxpArray(self.data[key], self.bits, self.exp, self.signed)

    def transpose(self) -> "FxpArray":
        return FxpArray(self.data.T, self.bits, self.exp, self.signed)

    @staticmethod
    def Zero(shape=(1,), bits=16, exp=8, signed=True):
        return FxpArray(torch.zeros(shape, dtype=torch.int32, device="cuda"), bits, exp, signed)

    @staticmethod
    def Zero_like(arr: "FxpArray"):
        return FxpArray.Zero(arr.data.shape, arr.bits, arr.exp, arr.signed)


@dataclass
class ComplexFxpArray:  # :170-229
    real: FxpArray
    imag: FxpArray

    @property
    def shape(self):
        assert self.real.shape == self.imag.shape, "real and imag shapes do not match"
        return self.real.shape

    @property
    def ndim(self):
        return self.real.ndim

    def copy(self) -> "ComplexFxpArray":
        return ComplexFxpArray(self.real.copy(), self.imag.copy())

    def clip(self, **kw):
        return ComplexFxpArray(self.real.clip(**kw), self.imag.clip(**kw))

    def to_float(self) -> torch.Tensor:
        return torch.complex(self.real.to_float(), self.imag.to_float())

    def transpose(self) -> "ComplexFxpArray":
        return ComplexFxpArray(self.real.transpose(), self.imag.transpose())

    def __getitem__(self, key) -> "ComplexFxpArray":
        return ComplexFxpArray(self.real[key], self.imag[key])


def _signed_only(*arrs: FxpArray):
    for a in arrs:
        if not a.signed:
            raise NotImplementedError("unsigned FxpArrays are not used by the fxp model and are not implemented")


def _floor_only(round_mode: RoundingMode, what: str):
    if round_mode != RoundingMode.FLOOR:
        raise NotImplementedError(f"{what}: only RoundingMode.FLOOR is implemented (the model uses no other)")


def fxp_minval(arr: FxpArray) -> int:  # :329-330
    return -(1 << (arr.bits - 1)) if arr.signed else 0


def fxp_maxval(arr: FxpArray) -> int:  # :333-334
    return (1 << (arr.bits - 1)) - 1 if arr.signed else (1 << arr.bits) - 1


def fxp_from_fp(x, bits: int = 16, exp: int = 8, signed: bool = True,
                round_mode: RoundingMode = RoundingMode.FLOOR, warn_on_clip: bool = True, warn_prefix: str = "",
                dtype=None) -> FxpArray:
    """:287-307."""
    if not signed:
        raise NotImplementedError("unsigned conversion is not implemented")
    if round_mode == RoundingMode.STOCHASTIC:
        raise NotImplementedError(f"rounding mode '{round_mode}' not implemented")
    xf = _dev(x, torch.float32)
    y = torch.empty(xf.shape, dtype=torch.int32, device=xf.device)
    check(lib.s5fxp_from_fp(_ptr(xf), _ptr(y), xf.numel(), bits, exp, round_mode.value, _stream()), "fxp_from_fp")
    return FxpArray(y, bits, exp, signed)


def fxp_change_cfg(x: FxpArray, new_bits: int, new_exp: int, new_signed: bool,
                   round_mode: RoundingMode = RoundingMode.FLOOR, warn_on_clip: bool = True,
                   warn_prefix: str = "") -> FxpArray:
    """:232-271."""
    if x.bits == new_bits and x.exp == new_exp and x.signed == new_signed:
        return x
    _signed_only(x)
    if not new_signed:
        raise NotImplementedError("unsigned FxpArrays are not implemented")
    _floor_only(round_mode, "fxp_change_cfg")
    src = x.data.contiguous()
    y = torch.empty_like(src)
    check(lib.s5fxp_change_cfg(_ptr(src), _ptr(y), src.numel(), x.bits, x.exp, new_bits, new_exp, _stream()),
          "fxp_change_cfg")
    return FxpArray(y, new_bits, new_exp, new_signed)


def fxp_change_exp(arr: FxpArray, new_exp: int, round_mode: RoundingMode = RoundingMode.FLOOR,
                   warn_on_clip: bool = True, warn_prefix: str = "") -> FxpArray:
    """:310-326 (no clip when the exponent is unchanged)."""
    if new_exp == arr.exp:
        return arr.copy()
    _signed_only(arr)
    _floor_only(round_mode, "fxp_change_exp")
    src = arr.data.contiguous()
    y = torch.empty_like(src)
    check(lib.s5fxp_change_cfg(_ptr(src), _ptr(y), src.numel(), arr.bits, arr.exp, arr.bits, new_exp, _stream()),
          "fxp_change_exp")
    return FxpArray(y, arr.bits, new_exp, arr.signed)


def fxp_clip(arr: FxpArray, do_warn: bool = False, warn_prefix: str = "") -> FxpArray:
    """:346-357.  The reference's overflow log line is not reproduced (it forces a host sync per op)."""
    _signed_only(arr)
    src = arr.data.contiguous()
    y = torch.empty_like(src)
    # change_cfg from a wider container to `bits` at the same exponent is exactly a clip
    check(lib.s5fxp_change_cfg(_ptr(src), _ptr(y), src.numel(), 32, arr.exp, arr.bits, arr.exp, _stream()),
          "fxp_clip")
    return FxpArray(y, arr.bits, arr.exp, arr.signed)


def _bcast_len(x: torch.Tensor, y: torch.Tensor) -> int:
    """Trailing-axis broadcast: y's shape must equal the last y.ndim axes of x."""
    if y.shape == x.shape:
        return x.numel()
    if y.ndim <= x.ndim and tuple(x.shape[x.ndim - y.ndim:]) == tuple(y.shape):
        return y.numel()
    raise NotImplementedError(f"unsupported broadcast {tuple(x.shape)} with {tuple(y.shape)}")


def _read_exp(dev: torch.Tensor, what: str) -> int:
    e, st = dev.tolist()  # one device->host sync, like the reference's int(np.ceil(...))
    if st & _lib.ST_NEGSHIFT:
        raise ValueError(f"invalid result_exp: {e}")  # :619-621
    if st & _lib.ST_NEGEXP:
        raise ValueError(f"{what}: negative result exponent {e}")
    return int(e)


def fxp_add(op1, op2, result_bits: Optional[int] = None, result_bits_fn: Callable[[int, int], int] = max,
            result_bits_add: int = 0, result_exp: Optional[Union[int, str]] = None, warn_on_overflow: bool = True,
            warn_on_neq_exp: bool = False, round_mode: RoundingMode = RoundingMode.FLOOR,
            warn_on_clip: bool = True, _negate_op2: bool = False) -> FxpArray:
    """:386-466."""
    if not (isinstance(op1, FxpArray) and isinstance(op2, FxpArray)):
        return TypeError(f"unsupported type(s) for fxp_add: '{type(op1)}' and '{type(op2)}'")
    _signed_only(op1, op2)
    _floor_only(round_mode, "fxp_add")
    if result_bits is None:
        result_bits = result_bits_fn(op1.bits, op2.bits) + result_bits_add
    a, b = op1.data.contiguous(), op2.data.contiguous()
    if b.numel() > a.numel():
        if _negate_op2:
            raise NotImplementedError("fxp_sub with a broadcast first operand")
        a, b, op1, op2 = b, a, op2, op1
    ylen = _bcast_len(a, b)
    out = torch.empty_like(a)
    if result_exp is None:
        if op1.exp != op2.exp:
            # :414-419 -- operator precedence makes that branch a different function; the model never
            # reaches it (every call passes result_exp)
            raise NotImplementedError("fxp_add with unequal exponents needs an explicit result_exp")
        result_exp = op1.exp
    if isinstance(result_exp, str):
        if result_exp != "compute_best":
            raise ValueError(f"invalid result_exp: {result_exp}")
        if _negate_op2:
            b = torch.neg(b)  # -1 * data, unclipped (:374); int32 negation wraps like JAX
        scratch = torch.empty(8, dtype=torch.int32, device=a.device)
        edev = torch.empty(2, dtype=torch.int32, device=a.device)
        check(lib.s5fxp_add_cb(_ptr(a), _ptr(b), _ptr(out), a.numel(), ylen, op1.bits, op1.exp, op2.bits, op2.exp,
                               result_bits, _ptr(edev), _ptr(scratch), _stream()), "fxp_add")
        return FxpArray(out, result_bits, _read_exp(edev, "fxp_add"), True)
    check(lib.s5fxp_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), ylen, op1.bits, op1.exp, op2.bits, op2.exp, result_bits,
                        int(result_exp), 1 if _negate_op2 else 0, _stream()), "fxp_add")
    return FxpArray(out, result_bits, int(result_exp), True)


def fxp_sub(op1, op2, result_bits: Optional[int] = None, result_bits_fn: Callable[[int, int], int] = max,
            result_bits_add: int = 0, result_exp: Optional[Union[int, str]] = None, warn_on_overflow: bool = True,
            warn_on_neq_exp: bool = False, round_mode: RoundingMode = RoundingMode.FLOOR,
            warn_on_clip: bool = True) -> FxpArray:
    """:360-383: fxp_add(op1, -1 * op2)."""
    return fxp_add(op1, op2, result_bits, result_bits_fn, result_bits_add, result_exp, warn_on_overflow,
                   warn_on_neq_exp, round_mode, warn_on_clip, _negate_op2=True)


def fxp_mul(op1, op2, result_exp: Optional[Union[int, str]] = None, result_exp_fn: Callable[[int, int], int] = max,
            result_bits: Optional[int] = None, result_bits_fn: Callable[[int, int], int] = max,
            round_mode: RoundingMode = RoundingMode.FLOOR, warn_on_overflow: bool = True) -> FxpArray:
    """:573-637."""
    if not (isinstance(op1, FxpArray) and isinstance(op2, FxpArray)):
        return TypeError(f"unsupported type(s) for fxp_mul: '{type(op1)}' and '{type(op2)}'")
    _signed_only(op1, op2)
    _floor_only(round_mode, "fxp_mul")
    if result_bits is None:
        result_bits = result_bits_fn(op1.bits, op2.bits)
    a, b = op1.data.contiguous(), op2.data.contiguous()
    if b.numel() > a.numel():
        a, b, op1, op2 = b, a, op2, op1
    ylen = _bcast_len(a, b)
    out = torch.empty_like(a)
    if result_exp is None:
        result_exp = result_exp_fn(op1.exp, op2.exp)
    if isinstance(result_exp, str):
        if result_exp != "compute_best":
            raise ValueError(f"invalid result_exp: {result_exp}")
        scratch = torch.empty(8, dtype=torch.int32, device=a.device)
        edev = torch.empty(2, dtype=torch.int32, device=a.device)
        check(lib.s5fxp_mul_cb(_ptr(a), _ptr(b), _ptr(out), a.numel(), ylen, op1.exp, op2.exp, result_bits, _ptr(edev),
                               _ptr(scratch), _stream()), "fxp_mul")
        return FxpArray(out, result_bits, _read_exp(edev, "fxp_mul"), True)
    if op1.exp + op2.exp - int(result_exp) < 0:
        raise ValueError(f"invalid result_exp: {result_exp}")  # :619-621
    check(lib.s5fxp_mul(_ptr(a), _ptr(b), _ptr(out), a.numel(), ylen, op1.exp, op2.exp, result_bits, int(result_exp),
                        _stream()), "fxp_mul")
    return FxpArray(out, result_bits, int(result_exp), True)


def fxp_matmul(op1, op2, result_bits: Optional[int] = None, result_bits_fn: Callable[[int, int], int] = max,
               result_exp: Optional[int] = None, result_exp_fn: Callable[[int, int], int] = max,
               round_mode: RoundingMode = RoundingMode.FLOOR) -> FxpArray:
    """:640-678.  op1: (..., K); op2: (K, M)."""
    if not (isinstance(op1, FxpArray) and isinstance(op2, FxpArray)):
        return TypeError(f"unsupported type(s) for fxp_matmul: '{type(op1)}' and '{type(op2)}'")
    _signed_only(op1, op2)
    _floor_only(round_mode, "fxp_matmul")
    if op2.ndim != 2 or op1.shape[-1] != op2.shape[0]:
        raise ValueError(f"fxp_matmul shapes {op1.shape} @ {op2.shape}")
    if result_bits is None:
        result_bits = result_bits_fn(op1.bits, op2.bits)
    if result_exp is None:
        result_exp = result_exp_fn(op1.exp, op2.exp)
    x, w = op1.data.contiguous(), op2.data.contiguous()
    K, M = w.shape
    N = x.numel() // K
    y = torch.empty(tuple(x.shape[:-1]) + (M,), dtype=torch.int32, device=x.device)
    check(lib.s5fxp_dense(_ptr(x), _ptr(w), None, _ptr(y), N, K, M, op1.exp, op2.exp, 0, 0, result_bits, int(result_exp), 0,
                          _stream()), "fxp_matmul")
    return FxpArray(y, result_bits, int(result_exp), True)


class CsrWeight:
    """A pruned (K,M) kernel stored by output channel (CSR of kernel^T) on the device, for ``fxp_matmul_csr``."""

    def __init__(self, w: np.ndarray, bits: int, exp: int, device=None):
        w = np.asarray(w, dtype=np.int32)
        self.K, self.M, self.bits, self.exp = int(w.shape[0]), int(w.shape[1]), int(bits), int(exp)
        wt = w.T
        nz = wt != 0
        self.nnz = int(nz.sum())
        rowptr = np.zeros(self.M + 1, dtype=np.int32)
        rowptr[1:] = np.cumsum(nz.sum(axis=1))
        cols = np.nonzero(nz)[1].astype(np.int32)
        vals = wt[nz].astype(np.int32)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rowptr = torch.from_numpy(rowptr).to(dev)
        self.colidx = torch.from_numpy(cols if self.nnz else np.zeros(1, np.int32)).to(dev)
        self.val = torch.from_numpy(vals if self.nnz else np.zeros(1, np.int32)).to(dev)

    @property
    def density(self) -> float:
        return self.nnz / float(self.K * self.M)


def fxp_matmul_csr(op1: FxpArray, op2: CsrWeight, result_bits: int, result_exp: int) -> FxpArray:
    """fxp_matmul (:640-678) with a pruned kernel: bit-identical to the dense op on the zero-filled kernel."""
    _signed_only(op1)
    if op1.shape[-1] != op2.K:
        raise ValueError(f"fxp_matmul_csr shapes {op1.shape} @ ({op2.K}, {op2.M})")
    x = op1.data.contiguous()
    N = x.numel() // op2.K
    y = torch.empty(tuple(x.shape[:-1]) + (op2.M,), dtype=torch.int32, device=x.device)
    check(lib.s5fxp_dense_csr(_ptr(x), _ptr(op2.rowptr), _ptr(op2.colidx), _ptr(op2.val), None, _ptr(y), N, op2.K, op2.M,
                              op1.exp, op2.exp, 0, 0, result_bits, int(result_exp), 0, _stream()), "fxp_matmul_csr")
    return FxpArray(y, result_bits, int(result_exp), True)


def fxp_isvalid(arr: FxpArray, do_warn: bool = False) -> bool:  # :704-721
    if arr.data is None or arr.bits <= 0 or arr.exp < 0:
        return False
    return bool(((arr.data >= arr.minval()) & (arr.data <= arr.maxval())).all())
