"""The FLOAT model's S5 SSM on MI355X: host-side mirror of ``sparseRNNs/model/ssm.py`` for inference.

Same names and argument meaning as the reference (``discretize_zoh`` ssm.py:37-50, ``binary_operator`` :54-77,
``apply_ssm`` = the ``_apply_ssm`` closure of :84-185 with identity quantisers, i.e. the un-quantised float model of
BASELINE configs[0]).  The scan over ``(Lambda_elements, Bu_elements)`` -- ``jax.lax.associative_scan`` at :127 and its
``reverse=True`` twin at :166-168 -- is one launch of the time-parallel HIP kernel behind ``s5fxp_assoc_scan_c64``
(``csrc/scan_assoc.hpp``); the B and C projections are plain complex / real GEMMs and go to the library
(``torch.matmul``: plumbing, as the task statement allows for plain library GEMMs).  There is no CPU path.

This is NOT the integer path: floating-point prefix sums depend on the combination tree, so results agree with the
reference's to rounding (tests state the bound), whereas ``fxpmodel.py`` is reproduced bit for bit.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from ._lib import check, lib


def _c64(x) -> torch.Tensor:
    if not torch.cuda.is_available():
        raise RuntimeError("sparsernns_amd needs a ROCm GPU: there is no CPU implementation of the SSM scan")
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    return x.to(device="cuda", dtype=torch.complex64).contiguous()


def discretize_zoh(Lambda, B_tilde, Delta):
    """ssm.py:37-50.  Lambda (P,) complex, B_tilde (P,H) complex, Delta (P,) float -> Lambda_bar (P,), B_bar (P,H)."""
    Lambda, B_tilde = _c64(Lambda), _c64(B_tilde)
    Delta = torch.as_tensor(np.asarray(Delta) if not isinstance(Delta, torch.Tensor) else Delta).to("cuda", torch.float32)
    Lambda_bar = torch.exp(Lambda * Delta)
    B_bar = (1 / Lambda * (Lambda_bar - 1))[..., None] * B_tilde
    return Lambda_bar, B_bar


def associative_scan(Lambda_bar, Bu_elements, reverse: bool = False, x0=None, return_last: bool = False):
    """``jax.lax.associative_scan(binary_operator, (Lambda_elements, Bu_elements), reverse=reverse)[1]`` (ssm.py:127,
    :166-168) for a time-invariant ``Lambda_elements = Lambda_bar * ones((L, P))`` (:106-108).

    Bu_elements: (L,P) or (B,L,P) complex64 on the GPU; returns xs of the same shape.  ``x0`` (B,P) / ``return_last``
    expose the carry for chunked (streaming) use; the reference's scan has neither (zeros, discarded)."""
    lam = _c64(Lambda_bar)
    bu = _c64(Bu_elements)
    squeeze = bu.dim() == 2
    if squeeze:
        bu = bu[None]
    if bu.dim() != 3 or lam.dim() != 1 or lam.shape[0] != bu.shape[2]:
        raise ValueError(f"associative_scan: Lambda_bar {tuple(lam.shape)} does not match Bu_elements {tuple(bu.shape)}")
    B, L, P = bu.shape
    xs = torch.empty_like(bu)
    x0t = None if x0 is None else _c64(x0).reshape(B, P)
    last = torch.empty((B, P), dtype=torch.complex64, device=bu.device) if return_last else None
    check(lib.s5fxp_assoc_scan_c64(lam.data_ptr(), bu.data_ptr(), xs.data_ptr(), None if x0t is None else x0t.data_ptr(),
                                   None if last is None else last.data_ptr(), B, L, P, 1 if reverse else 0,
                                   torch.cuda.current_stream().cuda_stream), "s5fxp_assoc_scan_c64")
    xs = xs[0] if squeeze else xs
    return (xs, last) if return_last else xs


def complex_relu(xs: torch.Tensor) -> torch.Tensor:
    """``jax.nn.relu`` on complex64 (ssm.py:161): maximum(x, 0) in lexicographic (re, im) order."""
    keep = (xs.real > 0) | ((xs.real == 0) & (xs.imag > 0))
    return torch.where(keep, xs, torch.zeros_like(xs))


def apply_ssm(Lambda_bar, B_bar, C_tilde, input_sequence, conj_sym: bool, bidirectional: bool, relufication: bool = False,
              B_bias=None, topk: float = 1.0, approx_topk: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """ssm.py:84-185 (``_apply_ssm``), un-quantised.  input_sequence (L,H) -- or (B,L,H): the reference vmaps the
    function over the batch -- float32; returns (ys (…,L,H) float32, xs (…,L,P or 2P) complex64)."""
    if topk < 1.0:
        raise NotImplementedError("Top-k sparsity is not part of the inference path (ssm.py:154-159)")
    lam, Bb, Ct = _c64(Lambda_bar), _c64(B_bar), _c64(C_tilde)
    u = input_sequence if isinstance(input_sequence, torch.Tensor) else torch.as_tensor(np.asarray(input_sequence))
    u = u.to(device="cuda", dtype=torch.float32)
    Bu = torch.matmul(u.to(torch.complex64), Bb.transpose(0, 1))            # b_dot, :110-119
    if B_bias is not None:
        Bu = Bu + _c64(B_bias)
    xs = associative_scan(lam, Bu)                                           # :127
    if relufication:
        xs = complex_relu(xs)                                                # :160-161
    if bidirectional:
        xs2 = associative_scan(lam, Bu, reverse=True)                        # :166-168
        xs = torch.cat((xs, xs2), dim=-1)                                    # :179
    ys = torch.matmul(xs.real, Ct.real.transpose(0, 1)) - torch.matmul(xs.imag, Ct.imag.transpose(0, 1))  # c_dot_real, :181-182
    return (2 * ys if conj_sym else ys), xs
