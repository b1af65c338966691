"""The float model's SSM scan (sparseRNNs/model/ssm.py:54-77,84-185).

Floating point: the tolerance is stated here.  A prefix sum's rounding depends on its combination tree (the reference's is
jax.lax.associative_scan's, ours is segment folds + sweeps over the segment aggregates), so the bar is: every state within
TOL * max|x| of the float64 sequential recurrence, and no further from it than a few times the error of the oracle's
complex64 restatement of the reference's own tree.  PARITY UNPINNED (no JAX here, no reference fixtures for this path).
"""
import numpy as np
import pytest

from oracle import float_ssm as F

TOL = 1e-5  # |x_gpu - x_f64| <= TOL * max|x_f64| (complex64 has 24-bit mantissas; 4096 steps with |lambda| < 1)


def _problem(B, L, P, seed, rho=(0.90, 0.9995)):
    rng = np.random.default_rng(seed)
    lam = (rng.uniform(*rho, P) * np.exp(1j * rng.uniform(-np.pi, np.pi, P))).astype(np.complex64)
    bu = (rng.standard_normal((B, L, P)) + 1j * rng.standard_normal((B, L, P))).astype(np.complex64)
    return lam, bu


@pytest.mark.parametrize("L", [1, 2, 3, 17, 64, 257])
def test_oracle_tree_agrees_with_the_sequential_recurrence(L):
    """CPU: the restated associative_scan (both directions) against float64 sequential; binary_operator is associative up
    to rounding."""
    lam, bu = _problem(1, L, 6, seed=L)
    A = np.broadcast_to(lam, (L, 6))
    for rev in (False, True):
        _, xs = F.associative_scan((A, bu[0]), reverse=rev)
        ref = F.scan_sequential_f64(lam, bu[0], reverse=rev)
        assert xs.dtype == np.complex64 and np.abs(xs - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())
    a, b, c = [(lam, bu[0, min(i, L - 1)]) for i in range(3)]
    l = F.binary_operator(F.binary_operator(a, b), c)
    r = F.binary_operator(a, F.binary_operator(b, c))
    assert np.allclose(l[0], r[0], rtol=1e-5) and np.allclose(l[1], r[1], rtol=1e-5, atol=1e-5)


def test_oracle_apply_ssm_shapes_and_conj_sym():
    rng = np.random.default_rng(3)
    L, H, P = 40, 6, 4
    lam, _ = _problem(1, L, P, 5)
    Bb = (rng.standard_normal((P, H)) + 1j * rng.standard_normal((P, H))).astype(np.complex64)
    u = rng.standard_normal((L, H)).astype(np.float32)
    Cc = (rng.standard_normal((H, P)) + 1j * rng.standard_normal((H, P))).astype(np.complex64)
    y1, xs1 = F.apply_ssm(lam, Bb, Cc, u, conj_sym=False, bidirectional=False)
    y2, _ = F.apply_ssm(lam, Bb, Cc, u, conj_sym=True, bidirectional=False)
    assert y1.shape == (L, H) and xs1.shape == (L, P) and np.allclose(y2, 2 * y1)
    C2 = np.concatenate([Cc, Cc], axis=1)
    y3, xs3 = F.apply_ssm(lam, Bb, C2, u, conj_sym=False, bidirectional=True, relufication=True)
    assert xs3.shape == (L, 2 * P) and y3.shape == (L, H)
    f = xs3[:, :P]   # the ReLU precedes the concatenation (ssm.py:160-179): only the forward half is rectified
    assert np.all((f.real > 0) | ((f.real == 0) & (f.imag >= 0))) and (xs3[:, P:].real < 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,P", [(1, 1, 1), (2, 15, 5), (1, 16, 16), (3, 511, 20), (2, 512, 64), (2, 513, 33), (1, 1024, 64),
                                   (2, 4096, 64), (1, 3000, 128)])
@pytest.mark.parametrize("reverse", [False, True])
def test_device_scan_within_tolerance_of_f64_and_of_the_reference_tree(B, L, P, reverse):
    import torch
    from sparsernns_amd import ssm

    lam, bu = _problem(B, L, P, seed=B * 1000 + L + P)
    xs = ssm.associative_scan(torch.as_tensor(lam), torch.as_tensor(bu), reverse=reverse).cpu().numpy()
    ref = F.scan_sequential_f64(lam, bu, reverse=reverse)
    scale = np.abs(ref).max()
    err = np.abs(xs - ref).max()
    assert err <= TOL * scale, (err, scale)
    if L <= 1024:   # the oracle's restatement of the reference's tree: our error is of the same order
        A = np.broadcast_to(lam, (L, P))
        tree = np.stack([F.associative_scan((A, bu[b]), reverse=reverse)[1] for b in range(B)])
        assert err <= 4 * np.abs(tree - ref).max() + 1e-6 * scale


@pytest.mark.gpu
def test_device_scan_carry_in_and_out_chain_chunks():
    """x0 / x_last: two chunks with the carry equal one scan over the concatenation (to rounding), in both layouts of L."""
    import torch
    from sparsernns_amd import ssm

    lam, bu = _problem(2, 700, 48, seed=9)
    whole = ssm.associative_scan(lam, bu).cpu().numpy()
    a, last = ssm.associative_scan(lam, bu[:, :300], return_last=True)
    b, last2 = ssm.associative_scan(lam, bu[:, 300:], x0=last, return_last=True)
    got = np.concatenate([a.cpu().numpy(), b.cpu().numpy()], axis=1)
    scale = np.abs(whole).max()
    assert np.abs(got - whole).max() <= TOL * scale
    assert np.abs(last.cpu().numpy() - whole[:, 299]).max() <= TOL * scale
    assert np.abs(last2.cpu().numpy() - whole[:, -1]).max() <= TOL * scale
    with pytest.raises(ValueError):
        ssm.associative_scan(lam[:5], bu)


@pytest.mark.gpu
@pytest.mark.parametrize("bidirectional,relufication,conj_sym", [(False, False, True), (True, True, False), (False, True, True)])
def test_apply_ssm_matches_the_oracle_at_config0_shape(bidirectional, relufication, conj_sym):
    """BASELINE configs[0]: B=1, L=1024 float forward of one SSM (ssm.py:84-185), against the oracle's restatement."""
    import torch
    from sparsernns_amd import ssm

    rng = np.random.default_rng(21)
    L, H, P = 1024, 96, 64
    lam, _ = _problem(1, L, P, 2)
    Bb = ((rng.standard_normal((P, H)) + 1j * rng.standard_normal((P, H))) / np.sqrt(H)).astype(np.complex64)
    Cc = ((rng.standard_normal((H, 2 * P if bidirectional else P)) + 1j * rng.standard_normal((H, 2 * P if bidirectional else P)))
          / np.sqrt(P)).astype(np.complex64)
    u = rng.standard_normal((L, H)).astype(np.float32)
    ys, xs = ssm.apply_ssm(lam, Bb, Cc, u, conj_sym, bidirectional, relufication)
    ry, rx = F.apply_ssm(lam, Bb, Cc, u, conj_sym, bidirectional, relufication)
    sx, sy = np.abs(rx).max(), np.abs(ry).max()
    # a state within rounding of zero may fall on the other side of the ReLU: compare where the oracle's state is clear of it
    clear = (np.abs(rx.real) > 1e-4 * sx) | ~np.bool_(relufication)
    if bidirectional:
        clear[:, P:] = True   # the reverse half is not rectified
    assert np.abs(xs.cpu().numpy() - rx)[clear].max() <= 1e-5 * sx
    assert np.abs(ys.cpu().numpy() - ry).max() <= 1e-4 * sy
    # batched call == per-sequence calls (the reference vmaps _apply_ssm over the batch)
    ub = np.stack([u, u[::-1].copy()])
    yb, _ = ssm.apply_ssm(lam, Bb, Cc, ub, conj_sym, bidirectional, relufication)
    assert torch.allclose(yb[0], ys, rtol=1e-5, atol=1e-5 * sy)
