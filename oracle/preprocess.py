"""ORACLE - TEST INFRASTRUCTURE ONLY.  numpy restatement of reference odt.py:10-19
(preprocess_image): tf.image.resize default = bilinear, antialias=False, half-pixel centres
[EXTERNAL TF2 ResizeBilinear CPU kernel: in = (out+0.5)*scale-0.5; lower = max(floor(in),0);
upper = min(ceil(in), size-1); lerp = in - floor(in); value = top + (bottom-top)*y_lerp with
top = tl + (tr-tl)*x_lerp], all float32; then tf.cast(float32 -> uint8) truncates toward zero.
Parity unpinned (tensorflow is not installed and the reference holds no resized frame)."""
import numpy as np


def _axis(n_out, n_in):
    scale = np.float32(n_in) / np.float32(n_out)
    src = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    fl = np.floor(src)
    lo = np.maximum(fl.astype(np.int64), 0)
    hi = np.minimum(np.ceil(src).astype(np.int64), n_in - 1)
    return lo, hi, (src - fl).astype(np.float32)


def preprocess_image(frame, input_size, swap_rb=False):
    frame = np.asarray(frame, np.uint8)
    h, w = int(input_size[0]), int(input_size[1])
    H, W = frame.shape[:2]
    y0, y1, fy = _axis(h, H)
    x0, x1, fx = _axis(w, W)
    f = frame.astype(np.float32)
    fx, fy = fx[None, :, None], fy[:, None, None]
    tl, tr = f[y0][:, x0], f[y0][:, x1]
    bl, br = f[y1][:, x0], f[y1][:, x1]
    top = tl + (tr - tl) * fx
    bot = bl + (br - bl) * fx
    out = (top + (bot - top) * fy).astype(np.int32).astype(np.uint8)
    if swap_rb:
        out = out[..., ::-1]
    return out[np.newaxis]
