"""Quantisation parameters exactly as the reference's runtime derives them from a model's tensor scales.

The reference executes `tflite-runtime==2.14.0` (reference requirements.txt:381; invoked at odt.py:58-61), which on
x86-64 applies the XNNPACK delegate by default with signed 8-bit quantised operators enabled.  Per operator of the
EfficientDet-Lite graph the kernel that runs is therefore [EXTERNAL: published TFLite 2.14 / XNNPACK sources, not in
the reference tree]:

  CONV_2D, DEPTHWISE_CONV_2D   XNNPACK qs8-qc8w igemm/dwconv, fp32 requantisation
                               scale[c] = (s_in * s_w[c]) / s_out in float32; q = clamp(rne(float(acc) * scale[c]) + z_out)
  ADD                          XNNPACK qs8-vadd-minmax (integer): see xnn_qs8_add_params below
  MAX_POOL_2D                  XNNPACK s8-maxpool (exact)
  LOGISTIC                     XNNPACK x8-lut, table built in float32: see xnn_qs8_sigmoid_lut below
  DEQUANTIZE                   float32 s * (q - z) (one multiplication)
  QUANTIZE (uint8 -> int8), RESIZE_NEAREST_NEIGHBOR, RESHAPE, CONCATENATION, TFLite_Detection_PostProcess
                               TFLite builtin kernels (byte moves; the post-process decodes in double precision)

The writers of the model container (tools/make_model.py, vbt_amd/tflite_import.py) call these functions; the HIP
library and the CPU oracle each re-derive the same numbers from the tensor scales in C++ / C and refuse a container
whose stored parameters differ, so three independent statements have to agree bit for bit.
"""
import math

import numpy as np

F32 = np.float32


def _libm_expf():
    """glibc's expf (what XNNPACK calls when it builds the table); falls back to the float rounding of the exact exp."""
    import ctypes
    import ctypes.util
    try:
        m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        f = m.expf
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
        return lambda v: F32(f(float(v)))
    except (OSError, AttributeError):
        return lambda v: F32(math.exp(float(v)))


def conv_requant_scales(s_in, s_w, s_out):
    """XNNPACK xnn_create_convolution2d_nhwc_qs8_qc8w: requantization_scale[c] = input_scale * kernel_scale[c] /
    output_scale, evaluated left to right in float32."""
    s_w = np.asarray(s_w, F32)
    return ((F32(s_in) * s_w).astype(F32) / F32(s_out)).astype(F32)


def xnn_qs8_add_params(s_a, s_b, s_out, z_a, z_b):
    """XNNPACK xnn_create_add_nd_qs8 + xnn_init_qs8_add_minmax_*_params.
    Returns (bias, a_multiplier, b_multiplier, shift): q = clamp(((bias + a*a_mult + b*b_mult) >> shift) + z_out)
    with an arithmetic shift; `bias` carries the rounding constant (round half towards +infinity) and both input
    zero points."""
    a_os = F32(s_a) / F32(s_out)              # input1_output_scale, float32 division
    b_os = F32(s_b) / F32(s_out)
    for v in (a_os, b_os):
        if not (v >= F32(2.0 ** -10) and v < F32(2.0 ** 8)):
            raise ValueError(f"ADD input/output scale ratio {float(v)} outside [2^-10, 2^8) (XNNPACK refuses it)")
    mx = max(a_os, b_os)
    max_exp = (int(np.asarray(mx, F32).view(np.uint32)) >> 23) - 127
    shift = 20 - max_exp                      # in [12, 30]
    assert 12 <= shift <= 30
    # multipliers: lrintf(scale * 2^shift) (round to nearest even); the larger one lands in [2^20, 2^21)
    a_mult = int(np.rint(F32(a_os * F32(2.0 ** shift))))
    b_mult = int(np.rint(F32(b_os * F32(2.0 ** shift))))
    assert max(a_mult, b_mult) >= 2 ** 20 and max(a_mult, b_mult) < 2 ** 21 + 1
    rounding = 1 << (shift - 1)
    bias = rounding - a_mult * int(z_a) - b_mult * int(z_b)
    assert -2 ** 31 <= bias < 2 ** 31
    return bias, a_mult, b_mult, shift


def xnn_qs8_add(a, b, params, z_out, act_min, act_max):
    """Vector form of the qs8-vadd-minmax micro-kernel (int64 numpy arithmetic; the kernel's int32 never overflows)."""
    bias, am, bm, shift = params
    acc = bias + np.asarray(a, np.int64) * am + np.asarray(b, np.int64) * bm
    assert np.all(np.abs(acc) < 2 ** 31)
    t = acc >> shift                                    # arithmetic
    t = np.clip(t, -32768, 32767) + int(z_out)          # _mm_packs_epi32 + _mm_adds_epi16
    t = np.clip(np.clip(t, -32768, 32767), -128, 127)   # _mm_packs_epi16
    return np.clip(t, act_min, act_max).astype(np.int8)


def xnn_qs8_sigmoid_lut(s_in, z_in, s_out=1.0 / 256.0, z_out=-128):
    """XNNPACK xnn_create_sigmoid_nc_qs8: 256-entry table in float32, index = int8 value + 128.
    x = s_in * (i - z_in); y = lrintf(min(max(256 / (1 + expf(-x)), qmin - z_out), qmax - z_out)) + z_out."""
    if F32(s_out) != F32(1.0 / 256.0) or int(z_out) != -128:
        raise ValueError("int8 LOGISTIC output must be quantised with scale 1/256, zero point -128")
    i = np.arange(-128, 128, dtype=np.int32)
    x = (F32(s_in) * (i - int(z_in)).astype(F32)).astype(F32)
    expf = _libm_expf()
    e = np.asarray([expf(-v) for v in x], F32)
    y = (F32(256.0) / (F32(1.0) + e)).astype(F32)
    y = np.minimum(np.maximum(y, F32(-128 - z_out)), F32(127 - z_out))
    return (np.rint(y).astype(np.int32) + int(z_out)).astype(np.int8)


def postprocess_tables(s_cls, z_cls, s_box, z_box, y_scale=1.0, h_scale=1.0):
    """Tables of the tail LOGISTIC -> DEQUANTIZE -> TFLite_Detection_PostProcess (detection_postprocess.cc,
    DecodeCenterSizeBoxes), indexed by int8 value + 128:
      score f32[256]  dequantised LOGISTIC output ((lut + 128) / 256)
      box   f32[256]  dequantised box encoding, float32 s_box * (q - z_box)
      dq    f64[256]  (double)box / (double)y_scale          -> ycenter = (float)(dq * (double)anchor.h + (double)anchor.y)
      ex    f64[256]  exp((double)box / (double)h_scale)     -> half_h  = (float)(0.5 * ex * (double)anchor.h)
    (x uses the same tables: the importer requires x_scale == y_scale and w_scale == h_scale.)"""
    lq = xnn_qs8_sigmoid_lut(s_cls, z_cls).astype(np.int32)
    score = ((lq + 128).astype(F32) * F32(1.0 / 256.0)).astype(F32)
    q = np.arange(-128, 128, dtype=np.int32)
    box = (F32(s_box) * (q - int(z_box)).astype(F32)).astype(F32)
    ys, hs = float(F32(y_scale)), float(F32(h_scale))
    dq = np.asarray([float(v) / ys for v in box], np.float64)
    ex = np.asarray([math.exp(float(v) / hs) for v in box], np.float64)
    return score, box, dq, ex


def pack_postprocess_tables(s_cls, z_cls, s_box, z_box, y_scale=1.0, h_scale=1.0):
    """Blob layout read by csrc/detector.hip and oracle/detector.c: score f32[256] | box f32[256] | dq f64[256] |
    ex f64[256] | scales f32[4] (y, x, h, w)."""
    score, box, dq, ex = postprocess_tables(s_cls, z_cls, s_box, z_box, y_scale, h_scale)
    sc = np.asarray([y_scale, y_scale, h_scale, h_scale], F32)
    return np.frombuffer(score.tobytes() + box.tobytes() + dq.tobytes() + ex.tobytes() + sc.tobytes(), np.uint8).copy()


POST_TABLE_BYTES = 256 * 4 * 2 + 256 * 8 * 2 + 16
