"""oracle/float_ssm.py -- TEST INFRASTRUCTURE: CPU restatement of the FLOAT model's SSM (sparseRNNs/model/ssm.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.  PARITY UNPINNED: the reference
needs JAX (absent here) and holds no fixtures for this path; what is restated is
  * binary_operator                (ssm.py:54-77 with identity quantisers: (A_j A_i, A_j b_i + b_j), complex64)
  * jax.lax.associative_scan       (third-party, jax >= 0.5.0 per pyproject.toml:26; its published algorithm: combine
                                    adjacent pairs, scan the half-length sequence recursively, interleave)
  * _apply_ssm                     (ssm.py:84-185: Bu = B_bar u, scan, optional complex ReLU, bidirectional reverse scan,
                                    y = Re(C x) or 2 Re(C x) with conj_sym)
plus a float64 sequential recurrence as the ground truth both are measured against.
"""
from __future__ import annotations

import numpy as np

C64 = np.complex64


def binary_operator(q_i, q_j):
    """ssm.py:54-77 with qhad = plain products: element i followed by element j."""
    A_i, b_i = q_i
    A_j, b_j = q_j
    return (A_j * A_i).astype(C64), (A_j * b_i + b_j).astype(C64)


def associative_scan(elems, reverse: bool = False):
    """jax.lax.associative_scan(binary_operator, (A, b)) along axis 0, restated (odd/even recursion)."""
    A, b = elems
    if reverse:
        ra, rb = associative_scan((A[::-1], b[::-1]))
        return ra[::-1], rb[::-1]
    n = A.shape[0]
    if n < 2:
        return A.astype(C64), b.astype(C64)
    # combine adjacent pairs (2k, 2k+1)
    ra, rb = binary_operator((A[0:n - 1:2], b[0:n - 1:2]), (A[1:n:2], b[1:n:2]))
    oa, ob = associative_scan((ra, rb))                      # inclusive results at odd positions 1, 3, 5, ...
    if n % 2 == 0:
        ea, eb = binary_operator((oa[:-1], ob[:-1]), (A[2::2], b[2::2]))
    else:
        ea, eb = binary_operator((oa, ob), (A[2::2], b[2::2]))
    ea = np.concatenate([A[:1].astype(C64), ea])             # position 0 is the first element itself
    eb = np.concatenate([b[:1].astype(C64), eb])
    out_a = np.empty_like(A, dtype=C64)
    out_b = np.empty_like(b, dtype=C64)
    out_a[0::2], out_b[0::2] = ea, eb
    out_a[1::2], out_b[1::2] = oa, ob
    return out_a, out_b


def scan_sequential_f64(lam, bu, reverse: bool = False, x0=None):
    """Ground truth: x_t = lam * x_{t-1} + bu_t in complex128.  bu: (..., L, P) -> same shape."""
    bu = np.asarray(bu).astype(np.complex128)
    lam = np.asarray(lam).astype(np.complex128)
    xs = np.empty_like(bu)
    x = np.zeros(bu.shape[:-2] + bu.shape[-1:], dtype=np.complex128) if x0 is None else np.asarray(x0).astype(np.complex128)
    L = bu.shape[-2]
    order = range(L - 1, -1, -1) if reverse else range(L)
    for t in order:
        x = lam * x + bu[..., t, :]
        xs[..., t, :] = x
    return xs


def complex_relu(xs):
    """jax.nn.relu on complex64 = maximum(x, 0), lexicographic (re, then im)."""
    keep = (xs.real > 0) | ((xs.real == 0) & (xs.imag > 0))
    return np.where(keep, xs, 0).astype(xs.dtype)


def apply_ssm(Lambda_bar, B_bar, C_tilde, input_sequence, conj_sym, bidirectional, relufication=False, B_bias=None):
    """ssm.py:84-185 for one (L,H) sequence, identity quantisers.  Returns (ys (L,H) float32, xs)."""
    u = np.asarray(input_sequence, dtype=np.float32)
    L = u.shape[0]
    Bu = (u.astype(C64) @ B_bar.T.astype(C64)).astype(C64)
    if B_bias is not None:
        Bu = (Bu + B_bias).astype(C64)
    lam = np.broadcast_to(Lambda_bar.astype(C64), (L, Lambda_bar.shape[0]))
    _, xs = associative_scan((lam, Bu))
    if relufication:
        xs = complex_relu(xs)
    if bidirectional:
        _, xs2 = associative_scan((lam, Bu), reverse=True)
        xs = np.concatenate([xs, xs2], axis=-1)
    y = (xs.real @ C_tilde.real.T - xs.imag @ C_tilde.imag.T).astype(np.float32)
    return (2 * y if conj_sym else y), xs
