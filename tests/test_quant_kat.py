"""Known-answer tests for the arithmetic the reference's runtime (tflite-runtime 2.14 -> XNNPACK delegate on x86-64)
applies outside the convolutions: int8 ADD (qs8-vadd-minmax), the LOGISTIC table and the decode of
TFLite_Detection_PostProcess.  No reference fixture exists for these (no model / tensor in the reference tree), so the
vectors are hand-derivable from the published formulas and computed here by an INDEPENDENT implementation (exact
rationals, `fractions.Fraction`), then required of all three statements: vbt_amd/quant.py (the model writers), the C
oracle (oracle/detector.c) and - through the container check at model load - the HIP library (tests/test_gpu_*)."""
import ctypes
import math
import struct
from fractions import Fraction

import numpy as np
import pytest

from vbt_amd import quant


def f32(x):
    """nearest float32 (ties to even) of a Fraction / float, as an exact Fraction"""
    return Fraction(struct.unpack("<f", struct.pack("<f", float(x)))[0])


def f32_div(a, b):
    # float32 division = the exact quotient rounded once; via double it is innocuous (53 >= 2*24 + 2 bits)
    return f32(Fraction(a) / Fraction(b))


def ref_add_params(sa, sb, so, za, zb):
    a_os, b_os = f32_div(f32(sa), f32(so)), f32_div(f32(sb), f32(so))
    mx = max(a_os, b_os)
    e = math.floor(math.log2(mx))
    assert Fraction(2) ** e <= mx < Fraction(2) ** (e + 1)
    shift = 20 - e
    am, bm = round(a_os * 2 ** shift), round(b_os * 2 ** shift)      # Fraction.__round__: ties to even, like lrintf
    return (2 ** (shift - 1)) - am * za - bm * zb, am, bm, shift


def ref_add(a, b, params, zo, lo, hi):
    bias, am, bm, shift = params
    acc = bias + a * am + b * bm
    t = acc // (2 ** shift)                                           # arithmetic shift = floor division
    t = max(-32768, min(32767, t))
    t = max(-32768, min(32767, t + zo))
    t = max(-128, min(127, t))
    return max(lo, min(hi, t))


# ---- hand-derived parameter vectors -------------------------------------------------------------------------------
# 0.3f = 10066330 * 2^-25, 0.7f = 11744051 * 2^-24, 1.0f exact.  max ratio 0.7 in [2^-1, 1) -> shift 21:
#   a_mult = rne(10066330 / 16) = rne(629145.625) = 629146 ; b_mult = rne(11744051 / 8) = rne(1468006.375) = 1468006
#   bias = 2^20 - 629146*3 - 1468006*(-4) = 1048576 - 1887438 + 5872024 = 5033162
HAND_PARAMS = [
    ((0.5, 0.5, 0.5, 0, 0), (2 ** 19, 2 ** 20, 2 ** 20, 20)),                    # all ratios 1.0: q = a + b + zo exactly
    ((0.25, 0.25, 0.5, 0, 0), (2 ** 20, 2 ** 20, 2 ** 20, 21)),                  # both 0.5: q = floor((a + b + 1) / 2) + zo
    ((0.25, 0.5, 0.5, 0, 0), (2 ** 19, 2 ** 19, 2 ** 20, 20)),                   # 0.5 and 1.0: q = floor((a + 2b + 1) / 2) + zo
    ((0.3, 0.7, 1.0, 3, -4), (5033162, 629146, 1468006, 21)),
    ((1.0, 1.0, 1.0 / 256.0, 0, 0), None),                                      # ratio 256 = 2^8: refused by XNNPACK
    ((1.0, 2.0 ** 11, 2.0 ** 11, 0, 0), None),                                  # ratio 2^-11 < 2^-10: refused
]


@pytest.mark.parametrize("args,want", HAND_PARAMS)
def test_add_params_hand_vectors(oracle_lib, args, want):
    L = oracle_lib.lib()
    L.vbto_add_params.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = (ctypes.c_int32 * 4)()
    rc = L.vbto_add_params(*args, out)
    if want is None:
        assert rc != 0
        with pytest.raises(ValueError):
            quant.xnn_qs8_add_params(*args)
        return
    assert rc == 0 and tuple(out) == want                            # C oracle
    assert tuple(quant.xnn_qs8_add_params(*args)) == want            # python writer
    assert ref_add_params(*args) == want                             # independent rational implementation


def test_add_params_random_scales_three_way(oracle_lib):
    L = oracle_lib.lib()
    L.vbto_add_params.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(11)
    out = (ctypes.c_int32 * 4)()
    n = 0
    for _ in range(600):
        so = float(np.float32(10.0 ** rng.uniform(-3, 0)))
        sa = float(np.float32(so * 2.0 ** rng.uniform(-9.5, 7.5)))
        sb = float(np.float32(so * 2.0 ** rng.uniform(-9.5, 7.5)))
        za, zb = int(rng.integers(-128, 128)), int(rng.integers(-128, 128))
        want = ref_add_params(sa, sb, so, za, zb)
        assert L.vbto_add_params(sa, sb, so, za, zb, out) == 0
        assert tuple(out) == want == tuple(quant.xnn_qs8_add_params(sa, sb, so, za, zb))
        assert 12 <= want[3] <= 30 and 2 ** 20 <= max(want[1], want[2]) <= 2 ** 21
        n += 1
    assert n == 600


# ---- element kernel on ties ---------------------------------------------------------------------------------------
TIE_VECTORS = [
    # (params args, zo, lo, hi, [(a, b, q)...])  -- q worked out by hand from floor((bias + a*am + b*bm) / 2^shift) + zo
    ((0.25, 0.25, 0.5, 0, 0), 0, -128, 127, [(1, 0, 1), (0, 1, 1), (-1, 0, 0), (-2, -1, -1), (3, 0, 2), (127, 127, 127), (-128, -128, -128), (-128, -127, -127)]),
    ((0.25, 0.5, 0.5, 0, 0), -5, -128, 127, [(1, 0, -4), (-1, 0, -5), (-3, 1, -5), (-3, 2, -4), (1, 1, -3), (127, 127, 127), (-127, -128, -128)]),
    ((0.5, 0.5, 0.5, 0, 0), 7, -3, 90, [(3, 4, 14), (-100, 2, -3), (100, 27, 90), (50, 33, 90), (-5, -5, -3)]),       # fused activation range
]


@pytest.mark.parametrize("pargs,zo,lo,hi,cases", TIE_VECTORS)
def test_add_kernel_tie_vectors(oracle_lib, pargs, zo, lo, hi, cases):
    params = ref_add_params(*pargs)
    a = np.asarray([c[0] for c in cases], np.int8)
    b = np.asarray([c[1] for c in cases], np.int8)
    want = np.asarray([c[2] for c in cases], np.int8)
    assert [ref_add(int(x), int(y), params, zo, lo, hi) for x, y in zip(a, b)] == want.tolist()
    assert np.array_equal(quant.xnn_qs8_add(a, b, params, zo, lo, hi), want)
    L = oracle_lib.lib()
    L.vbto_add_vec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    q4 = (ctypes.c_int32 * 4)(*params)
    got = np.zeros(len(cases), np.int8)
    L.vbto_add_vec(a.ctypes.data, b.ctypes.data, len(cases), q4, zo, lo, hi, got.ctypes.data)
    assert np.array_equal(got, want)
    # the float form this build used before (rne of the real-valued sum) disagrees on the first vector of the 0.5/0.5 case
    if pargs == (0.25, 0.25, 0.5, 0, 0):
        assert int(np.rint(np.float32(0.5) * 1 + np.float32(0.5) * 0)) == 0 != want[0]


def test_add_kernel_exhaustive_against_rationals(oracle_lib):
    """all 65 536 (a, b) pairs for a few parameter sets: C oracle == numpy writer == exact rational statement"""
    L = oracle_lib.lib()
    L.vbto_add_vec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    a, b = [x.reshape(-1).astype(np.int8) for x in np.meshgrid(np.arange(-128, 128), np.arange(-128, 128), indexing="ij")]
    for pargs, zo, lo, hi in (((0.0431, 0.0517, 0.0622, -7, 12), 5, -128, 127), ((0.25, 0.5, 0.5, 1, -1), -128, -128, 40),
                              ((2.0, 0.004, 0.008, 100, -100), 0, -128, 127), ((0.001, 0.0009, 0.9, 0, 0), -20, -128, 127)):
        params = ref_add_params(*pargs)
        want = np.asarray([ref_add(int(x), int(y), params, zo, lo, hi) for x, y in zip(a[::37], b[::37])], np.int8)   # rationals: a sample
        got_np = quant.xnn_qs8_add(a, b, params, zo, lo, hi)
        got_c = np.zeros(a.size, np.int8)
        L.vbto_add_vec(a.ctypes.data, b.ctypes.data, a.size, (ctypes.c_int32 * 4)(*params), zo, lo, hi, got_c.ctypes.data)
        assert np.array_equal(got_np, got_c)
        assert np.array_equal(got_c[::37], want)


# ---- LOGISTIC table and decode ------------------------------------------------------------------------------------
def test_sigmoid_table_against_exact_sigmoid(oracle_lib):
    L = oracle_lib.lib()
    L.vbto_sigmoid_lut_entry.argtypes = [ctypes.c_float, ctypes.c_int, ctypes.c_int]
    for s_in, z_in in ((0.0625, 0), (0.09017, 11), (0.2, -30), (0.01, 100)):
        lut = quant.xnn_qs8_sigmoid_lut(s_in, z_in)
        assert lut.dtype == np.int8 and np.all(np.diff(lut.astype(int)) >= 0)
        far = 0
        for i in range(-128, 128):
            assert L.vbto_sigmoid_lut_entry(s_in, z_in, i) == int(lut[i + 128])
            x = float(np.float32(s_in)) * (i - z_in)
            exact = 256.0 / (1.0 + math.exp(-x))
            ideal = min(max(round(exact), 0), 255) - 128
            far += abs(int(lut[i + 128]) - ideal)
            assert abs(int(lut[i + 128]) - ideal) <= 1              # float32 evaluation: at most one step from the exact table
        assert far <= 2
    # hand values: x = 0 -> 256 / 2 = 128 -> q = 0 ; large |x| saturates to -128 / 127 (255 is the clamp, not 256)
    lut = quant.xnn_qs8_sigmoid_lut(0.0625, 0)
    assert lut[128] == 0 and lut[0] == -128 + round(256 / (1 + math.exp(8))) and lut[255] == min(127, round(256 / (1 + math.exp(-127 * 0.0625))) - 128)
    assert quant.xnn_qs8_sigmoid_lut(0.5, 0)[255] == 127 and quant.xnn_qs8_sigmoid_lut(0.5, 0)[0] == -128


def test_postprocess_tables_layout_and_decode_rule():
    s_box, z_box = np.float32(0.0213), 9
    score, box, dq, ex = quant.postprocess_tables(0.09, 10, s_box, z_box, y_scale=1.0, h_scale=1.0)
    assert score.dtype == box.dtype == np.float32 and dq.dtype == ex.dtype == np.float64
    assert np.all((score * 256) % 1 == 0) and score.min() >= 0 and score.max() <= 255 / 256           # reference score lattice k/256
    q = np.arange(-128, 128)
    assert np.array_equal(box, (s_box * (q - z_box).astype(np.float32)).astype(np.float32))           # DEQUANTIZE: one float32 product
    assert np.array_equal(dq, box.astype(np.float64))                                                 # scales 1.0
    assert ex.tolist() == [math.exp(float(v)) for v in box]            # libm exp (numpy's vectorised exp differs in the last bit)
    assert np.allclose(ex, np.exp(box.astype(np.float64)), rtol=4e-16, atol=0)
    blob = quant.pack_postprocess_tables(0.09, 10, s_box, z_box)
    assert blob.size == quant.POST_TABLE_BYTES == 6160
    # decode rule of detection_postprocess.cc on one anchor, by hand: double products, one rounding to float
    an = np.asarray([0.5125, 0.2875, 0.075, 0.15], np.float32)
    i = 40 + 128
    yc = np.float32(dq[i] * float(an[2]) + float(an[0]))
    hh = np.float32(0.5 * ex[i] * float(an[2]))
    assert yc == np.float32(float(box[i]) * float(an[2]) + float(an[0])) and hh > 0
    s2 = quant.postprocess_tables(0.09, 10, s_box, z_box, y_scale=10.0, h_scale=5.0)
    assert np.array_equal(s2[2], box.astype(np.float64) / 10.0) and s2[3].tolist() == [math.exp(float(v) / 5.0) for v in box]
