"""Synthetic clips: the reference's input videos are not in
its tree (hosted on GDrive, reference README.md:38-39), so throughput and parity runs use a
seeded stand-in: a static block-constant random background per clip plus two filled ellipses
("plates") of the median size seen in reference dfs_ocsort/001_* (w=0.28, h=0.16 normalised)
moving vertically like a squat rep.
"""
from __future__ import annotations

import numpy as np

PLATE_W = 0.28
PLATE_H = 0.16
PLATE_X = (0.35, 0.60)
FPS = 60.0
REP_PERIOD_S = 3.3


def plate_y(t_frame, fps: float = FPS):
    return 0.34 + 0.14 * np.sin(2.0 * np.pi * np.asarray(t_frame, dtype=np.float64) / (REP_PERIOD_S * fps))


def background(seed: int, size: int = 320) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    nb = -(-size // 8)
    blocks = rng.integers(0, 256, size=(nb, nb, 3), dtype=np.uint8)
    return np.repeat(np.repeat(blocks, 8, axis=0), 8, axis=1)[:size, :size].copy()


def render(bg: np.ndarray, t_frame: int, fps: float = FPS) -> np.ndarray:
    """One RGB uint8 frame [S,S,3] of the clip whose background is `bg` at frame index t."""
    size = bg.shape[0]
    img = bg.copy()
    cy = float(plate_y(t_frame, fps)) * size
    ry = PLATE_H * size / 2.0
    rx = PLATE_W * size / 2.0
    y0, y1 = max(int(cy - ry) - 1, 0), min(int(cy + ry) + 2, size)
    ys = np.arange(y0, y1, dtype=np.float64)[:, None] + 0.5
    for i, px in enumerate(PLATE_X):
        cx = px * size
        x0, x1 = max(int(cx - rx) - 1, 0), min(int(cx + rx) + 2, size)
        xs = np.arange(x0, x1, dtype=np.float64)[None, :] + 0.5
        d = ((ys - cy) / ry) ** 2 + ((xs - cx) / rx) ** 2
        sub = img[y0:y1, x0:x1]
        sub[d <= 1.0] = (36, 36, 44)
        sub[d <= 0.08] = (180, 180, 170)       # hub
    return img


def clip_frames(seed: int, t0: int, n: int, size: int = 320) -> np.ndarray:
    bg = background(seed, size)
    return np.stack([render(bg, t0 + i) for i in range(n)])


def batch_frames(seeds, t: int, size: int = 320) -> np.ndarray:
    """Frame t of each clip in `seeds`, batched frame-wise: [len(seeds), S, S, 3] uint8."""
    return np.stack([render(background(s, size), t) for s in seeds])
