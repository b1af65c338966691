"""Known-answer tests that pin the CPU oracle to values derived BY HAND from the reference's formulas.

The reference ships no tests or golden vectors for the fixed-point path and
JAX is not installable here, so these hand-derived answers -- each with the reference line it comes
from -- are what the oracle is anchored on ("parity unpinned" otherwise).
Citations: file:line into /root/reference/sparseRNNs/.
"""
import numpy as np
import pytest

from oracle import fxp_oracle as O

I32 = np.int32


def fx(v, bits=16, exp=0):
    return O.Fx(np.asarray(v, dtype=np.int64).astype(I32), bits, exp)


def test_sigmoid_lut_values():
    # fxpmodel.py:89-95: x_k = k * 2^x_exp, lut[k] = rint(sigmoid(k) * 2^y_exp) - 2^(y_exp-1)
    # sigmoid(k) for k=0..7: .5, .7310586, .8807971, .9525741, .9820138, .9933071, .9975274, .9990889
    assert O.sigmoid_lut(6, 14).tolist() == [0, 3786, 6239, 7415, 7897, 8082, 8151, 8177]
    assert O.sigmoid_lut(6, 6).tolist() == [0, 15, 24, 29, 31, 32, 32, 32]
    # the sample points are k*2^x_exp whatever x_exp is, so the table does not depend on x_exp
    assert O.sigmoid_lut(4, 14).tolist() == O.sigmoid_lut(6, 14).tolist()


def test_sigmoid_apply_by_hand():
    lut = O.sigmoid_lut(6, 14)
    # fxpmodel.py:109-119,139-140 with x_exp=6, y_exp=14: yy = 8192 + sgn * (((64-mu)*lut[i] >> 6) + (mu*lut[i+1] >> 6))
    x = fx([0, 64, 32, -32, 448, 449, 1000, -5], 16, 6)
    y = O.sigmoid_apply(x, 6, 14, lut)
    half = lambda a: ((64 - (a & 63)) * int(lut[min(a >> 6, 6)]) >> 6) + ((a & 63) * int(lut[min(a >> 6, 6) + 1]) >> 6)
    want = [8192 - 0, 8192 + 3786, 8192 + half(32), 8192 - half(32), 8192 + half(448), 8192 + half(449),
            8192 + half(1000), 8192 - half(5)]
    assert y.data.tolist() == want
    assert (y.bits, y.exp) == (16, 14)
    # xx == 0 takes the "negative" sign (2*(xx>0)-1 = -1) but half(0) = 0
    assert want[0] == 8192
    # saturation quirk: beyond 7*2^x_exp the index sticks at 6 while mu keeps wrapping
    assert half(448) == int(lut[6])
    assert half(511) > half(512) == int(lut[6])  # the output drops back to lut[6] every 64 steps
    assert O.sigmoid_apply(fx([511, 512], 16, 6), 6, 14, lut).data.tolist() == [8192 + half(511), 8192 + half(512)]
    # input exponent above x_exp is first shifted down with floor (change_exp, fxparray.py:321-325)
    y2 = O.sigmoid_apply(fx([-129, 129], 16, 7), 6, 14, lut)
    assert y2.data.tolist() == [8192 - half(65), 8192 + half(64)]  # -129>>1 = -65, 129>>1 = 64


def test_rshift_modes_and_change_exp_quirks():
    d = np.array([-5, -4, -1, 0, 1, 5, 7], dtype=I32)
    assert O.asr(d, 1, O.FLOOR).tolist() == [-3, -2, -1, 0, 0, 2, 3]  # fxparray.py:278
    assert O.asr(d, 1, O.CEIL).tolist() == [-2, -2, 0, 0, 1, 3, 4]  # :280
    assert O.asr(d, 1, O.ROUND).tolist() == [-2, -2, 0, 0, 1, 3, 4]  # :282 (x + 1) >> 1
    # unchanged exponent: NO clip even if data exceed the bits (fxparray.py:318-319)
    a = fx([70000, -70000], 16, 5)
    assert O.change_exp(a, 5).data.tolist() == [70000, -70000]
    # left shift saturates at the CURRENT bits (fxparray.py:321-325) ...
    assert O.change_exp(fx([20000, -20000, 3], 16, 5), 7).data.tolist() == [32767, -32768, 12]
    # ... even when change_cfg is about to widen the bits (fxparray.py:244-261)
    r = O.change_cfg(fx([20000], 16, 5), 20, 7)
    assert (r.data.tolist(), r.bits, r.exp) == ([32767], 20, 7)
    # shrinking bits clips at the new bits after the shift
    r = O.change_cfg(fx([20000, -3], 16, 5), 8, 3)
    assert r.data.tolist() == [127, -1]


def test_from_fp_rounding():
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.49, 1000.0, -1000.0], dtype=np.float32)
    assert O.from_fp(x, 8, 0, True, O.ROUND).data.tolist() == [0, 2, 2, 0, -2, 0, 127, -128]  # half-to-even, clip
    assert O.from_fp(x, 8, 0, True, O.FLOOR).data.tolist() == [0, 1, 2, -1, -2, 0, 127, -128]
    assert O.from_fp(x, 16, 4, True, O.FLOOR).data.tolist() == [8, 24, 40, -8, -24, 7, 16000, -16000]


def test_add_numeric_and_sub():
    # fxparray.py:449-464: each operand -> change_exp (clip at own bits) -> add -> clip at result bits
    a, b = fx([100, 30000, -30000], 16, 4), fx([3, 30000, -30000], 16, 6)
    r = O.add(a, b, 16, 6)
    # a<<2 = [400, 120000->32767, -120000->-32768]; sums [403, 62767->32767, -62768->-32768]
    assert r.data.tolist() == [403, 32767, -32768]
    r = O.add(a, b, 16, 4)  # b>>2 floor = [0, 7500, -7500]
    assert r.data.tolist() == [100, 32767, -32768]
    # sub = add(a, -1*b) with the negation unclipped (fxparray.py:374): -(-32768) = 32768 enters the sum
    r = O.sub(fx([0, -1], 16, 0), fx([-32768, -32768], 16, 0), 16, 0)
    assert r.data.tolist() == [32767, 32767]


def test_mul_and_matmul_wrap():
    # fxparray.py:623-626: int32 product wraps before the shift
    a, b = fx([65536, 3, -7], 32, 2), fx([65536, 5, 3], 32, 3)
    r = O.mul(a, b, 32, 5)  # rshift 0: 2^32 wraps to 0
    assert r.data.tolist() == [0, 15, -21]
    r = O.mul(fx([-7], 16, 2), fx([3], 16, 3), 16, 3)  # -21 >> 2 = -6 (floor)
    assert r.data.tolist() == [-6]
    with pytest.raises(ValueError):
        O.mul(a, b, 32, 6)  # negative shift, fxparray.py:619-621
    # matmul: int32 accumulate with wrap (fxparray.py:662), then floor shift and clip
    x = fx([[1 << 30, 1 << 30, 5]], 32, 0)
    w = fx([[2], [2], [1]], 32, 0)
    assert O.matmul(x, w, 32, 0).data.tolist() == [[5]]  # 2^32 wraps away
    assert O.matmul(fx([[7, -9]], 16, 3), fx([[3], [2]], 8, 2), 16, 4).data.tolist() == [[1]]  # (21-18)=3 >> 1


def test_compute_best_add_by_hand():
    # fxparray.py:420-448.  F(a) = [1.5, -3.25], F(b) = [2.0, 0.25]; max|F(a)+F(b)| = 3.5 -> intbits 2 -> exp 13
    a, b = fx([24, -52], 16, 4), fx([128, 16], 16, 6)
    r = O.add(a, b, None, "compute_best")
    assert (r.bits, r.exp) == (16, 13)
    # agg_exp = 6; a<<2 = [96,-208]; sums [224,-192]; exp 6 -> 13: <<7
    assert r.data.tolist() == [224 << 7, -(192 << 7)]
    # a maximum exactly on a power of two: log2(4 + 1e-6) > 2 -> 3 integer bits
    r = O.add(fx([64], 16, 4), fx([0], 16, 4), None, "compute_best")
    assert r.exp == 16 - 3 - 1
    r = O.add(fx([63], 16, 4), fx([0], 16, 4), None, "compute_best")  # 3.9375 -> 2 integer bits
    assert r.exp == 16 - 2 - 1
    # all-zero tensors: log2(1e-6) < 0 -> intbits 0 -> exp = bits - 1
    assert O.add(fx([0, 0], 16, 4), fx([0, 0], 16, 9), None, "compute_best").exp == 15


def test_compute_best_add_saturation_quirk():
    # widening an operand to agg_exp saturates it at its OLD 16 bits
    z = fx([26000], 16, 12)  # 6.35
    s = fx([12000], 16, 15)  # 0.366
    r = O.add(z, s, 16, "compute_best")
    # max|f| = 6.71 -> intbits 3 -> exp 12; z<<3 clips to 32767; (32767 + 12000) >> 3 = 5595
    assert (r.exp, r.data.tolist()) == (12, [5595])


def test_compute_best_mul_by_hand():
    # fxparray.py:601-609,619-626: F = [1.5*0.75] = 1.125 -> intbits 1 -> exp 14; rshift = 4+6-14 < 0 -> ValueError
    with pytest.raises(ValueError):
        O.mul(fx([24], 16, 4), fx([48], 16, 6), None, "compute_best")
    r = O.mul(fx([24 << 8], 16, 12), fx([48 << 4], 16, 10), None, "compute_best")  # same values, finer inputs
    assert (r.exp, r.data.tolist()) == (14, [(24 << 8) * (48 << 4) >> 8])


def test_complex_relu_truth_table():
    # fxpmodel.py:30-45: lexicographic maximum(z, 0): keep iff re > 0 or (re == 0 and im > 0)
    re = fx([5, 0, 0, 0, -3, 7])
    im = fx([-9, 4, 0, -4, 8, 0])
    r, i = O.complex_relu(re, im)
    assert r.data.tolist() == [5, 0, 0, 0, 0, 7]
    assert i.data.tolist() == [-9, 4, 0, 0, 0, 0]
    # values pass through float32: 2^24+1 is not representable
    r, i = O.complex_relu(fx([2**24 + 1], 32), fx([-(2**24) - 3], 32))
    assert (r.data.tolist(), i.data.tolist()) == ([2**24], [-(2**24) - 4])
    assert O.relu(fx([-2, 0, 3])).data.tolist() == [0, 0, 3]


def test_scan_by_hand():
    # fxpmodel.py:155-169 with Lambda = 0.5 (A_re = 16384 @ exp 15, A_im = 0): x_t = (x_{t-1} >> 1) + Bu_t, floor on negatives
    bu = fx(np.array([[[-7], [0], [0], [10], [-1]]]), 16, 14)
    zero = fx(np.zeros((1, 5, 1)), 16, 14)
    xr, xi = O.scan(bu, zero, fx([16384], 16, 15), fx([0], 16, 15), 14, 14)
    assert xr[0, :, 0].tolist() == [-7, -4, -2, 9, 3]  # -7>>1=-4, -4>>1=-2, -2>>1=-1 (+10), 9>>1=4 (-1)
    assert xi[0, :, 0].tolist() == [0, 0, 0, 0, 0]
    # pure rotation by i (A_re = 0, A_im = 2^14 @ exp 14): re' = -im + bre, im' = re + bim
    bre = fx(np.array([[[3], [0], [0], [0]]]), 16, 5)
    xr, xi = O.scan(bre, fx(np.zeros((1, 4, 1)), 16, 5), fx([0], 16, 14), fx([1 << 14], 16, 14), 5, 5)
    assert (xr[0, :, 0].tolist(), xi[0, :, 0].tolist()) == ([3, 0, -3, 0], [0, 3, 0, -3])
    # Bu exponent above / below the state exponent: shift right (floor) / left (fxpmodel.py:158-167)
    xr, _ = O.scan(fx(np.array([[[-5]]]), 16, 6), fx(np.zeros((1, 1, 1)), 16, 6), fx([0], 16, 15), fx([0], 16, 15), 4, 4)
    assert xr.tolist() == [[[-2]]]
    xr, _ = O.scan(fx(np.array([[[-5]]]), 16, 4), fx(np.zeros((1, 1, 1)), 16, 4), fx([0], 16, 15), fx([0], 16, 15), 6, 6)
    assert xr.tolist() == [[[-20]]]


def test_scan_wraps_and_never_clips():
    # the state is never saturated (fxpmodel.py:147-172) and A*x wraps at 32 bits
    bu = fx(np.array([[[2**30], [0]]]), 32, 0)
    xr, _ = O.scan(bu, fx(np.zeros((1, 2, 1)), 32, 0), fx([4], 16, 0), fx([0], 16, 0), 0, 0)
    assert xr[0, :, 0].tolist() == [2**30, 0]  # 4 * 2^30 = 2^32 -> wraps to 0
    xr, _ = O.scan(fx(np.array([[[40000], [40000]]]), 32, 0), fx(np.zeros((1, 2, 1)), 32, 0), fx([1], 16, 0), fx([0], 16, 0), 0, 0)
    assert xr[0, :, 0].tolist() == [40000, 80000]  # beyond 16 bits, untouched


def test_ceil_log2_definition():
    # the oracle's documented definition: correctly rounded float32 log2, then ceil
    assert O.ceil_log2_f32(np.float32(1.0)) == 0
    assert O.ceil_log2_f32(np.float32(2.0)) == 1
    assert O.ceil_log2_f32(np.nextafter(np.float32(2.0), np.float32(3.0))) == 2
    assert O.ceil_log2_f32(np.float32(0.3)) == -1
    # a few ulps above 2^5 the float32 log2 still rounds to 5.0 (documented ambiguity zone)
    v = np.float32(32.0)
    for _ in range(3):
        v = np.nextafter(v, np.float32(64.0))
    assert O.ceil_log2_f32(v) in (5, 6)
    assert O.intbits_f32(np.float32(0.0), 1e-6) == 0
