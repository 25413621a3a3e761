"""oracle/preprocess.py (reference odt.py:10-19: tf.image.resize bilinear, half-pixel centres, no antialias, then a
truncating uint8 cast) cross-checked against an INDEPENDENT implementation of the same published formula: torch's CPU
`interpolate(mode="bilinear", align_corners=False, antialias=False)`.  TensorFlow itself is absent, so this does not pin
the oracle to TF bit for bit - it pins the sampling geometry (source coordinate (dst + 0.5) * in/out - 0.5, edge clamp,
two-tap lerp): the two float32 evaluation orders differ by at most one ulp, which the truncating cast turns into a
difference of exactly 1 on the rare pixel whose interpolated value sits on an integer."""
import numpy as np
import pytest


@pytest.mark.parametrize("src,dst", [((1920, 1080), (320, 320)), ((720, 1280), (448, 448)), ((7, 5), (320, 320)), ((320, 320), (320, 320)),
                                     ((333, 517), (384, 384)), ((64, 48), (20, 31))])
def test_bilinear_geometry_matches_an_independent_implementation(src, dst):
    import torch
    from oracle.preprocess import preprocess_image
    rng = np.random.default_rng(src[0] * 7 + dst[0])
    img = rng.integers(0, 256, (src[0], src[1], 3), dtype=np.uint8)
    got = preprocess_image(img, dst)[0].astype(np.int32)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].to(torch.float32)
    ref = torch.nn.functional.interpolate(t, size=dst, mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0)
    ref = ref.numpy().astype(np.int32)                      # numpy's float -> int cast truncates toward zero like tf.cast
    diff = np.abs(got - ref)
    assert diff.max() <= 1, diff.max()
    assert (diff != 0).mean() < 2e-3, (diff != 0).mean()     # only values that sit on an integer may flip


def test_identity_and_constant_images():
    from oracle.preprocess import preprocess_image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(preprocess_image(img, (37, 53))[0], img)           # same size: every sample lands on a pixel centre
    flat = np.full((90, 41, 3), 173, np.uint8)
    assert np.array_equal(preprocess_image(flat, (320, 320))[0], np.full((320, 320, 3), 173, np.uint8))
    assert np.array_equal(preprocess_image(img, (64, 64), swap_rb=True)[0], preprocess_image(img[..., ::-1], (64, 64))[0])
