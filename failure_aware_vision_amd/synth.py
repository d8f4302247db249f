"""Deterministic synthetic frames for benchmarks, fixtures and calibration.

There is no dataset in the reference and no network here, so
frames are generated.  i.i.d. noise frames are useless for a label-parity test
(every frame has the same statistics, hence the same label), so each frame is a
random smooth colour field: a per-frame base colour plus a few random 2-D
cosine waves, quantised to uint8 like a camera frame
(platform/backend/video_source.py:144-148 hands the scorer uint8 HxWx3).

Every frame depends only on (seed, global frame index), so any rank can
generate exactly its own shard.
"""
from __future__ import annotations

import numpy as np

# ImageNet-C "gaussian_noise" sigmas for severity 1..5 on [0,1] pixels
# (external convention, SURVEY.md §8a "decisions"; not in the reference).
GAUSSIAN_NOISE_SIGMA = (0.08, 0.12, 0.18, 0.26, 0.38)


def synthetic_frame_u8(h: int, w: int, seed: int, frame_id: int) -> np.ndarray:
    """One uint8 [h, w, 3] frame, a pure function of (h, w, seed, frame_id)."""
    rng = np.random.default_rng([int(seed), int(frame_id)])
    nwave = 4
    base = rng.uniform(0.2, 0.8, size=3)
    amp = rng.uniform(-0.25, 0.25, size=(nwave, 3))
    fx = rng.uniform(-6.0, 6.0, size=nwave)
    fy = rng.uniform(-6.0, 6.0, size=nwave)
    ph = rng.uniform(0.0, 2.0 * np.pi, size=nwave)
    ys = (np.arange(h, dtype=np.float64) + 0.5) / h
    xs = (np.arange(w, dtype=np.float64) + 0.5) / w
    img = np.empty((h, w, 3), np.float64)
    img[:] = base
    for k in range(nwave):
        ay = 2.0 * np.pi * fy[k] * ys + ph[k]
        ax = 2.0 * np.pi * fx[k] * xs
        # cos(ax + ay) by the addition formula: two outer products, no HxW trig
        field = np.outer(np.cos(ay), np.cos(ax)) - np.outer(np.sin(ay), np.sin(ax))
        img += field[:, :, None] * amp[k][None, None, :]
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def synthetic_frames_u8(n: int, h: int, w: int, seed: int, start_id: int = 0) -> np.ndarray:
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        out[i] = synthetic_frame_u8(h, w, seed, start_id + i)
    return out


def gaussian_noise_f32(frames_u8: np.ndarray, severity: int, seed: int, start_id: int = 0) -> np.ndarray:
    """uint8 frames -> fp32 [0,1] frames with ImageNet-C style Gaussian noise,
    clipped.  Noise for frame i depends only on (seed, start_id + i)."""
    sigma = np.float32(GAUSSIAN_NOISE_SIGMA[severity - 1])
    out = np.empty(frames_u8.shape, np.float32)
    for i in range(frames_u8.shape[0]):
        rng = np.random.default_rng([int(seed), int(start_id + i), 0x6E6F6973])
        x = frames_u8[i].astype(np.float32) * np.float32(1.0 / 255.0)
        n = rng.standard_normal(x.shape, dtype=np.float32)
        out[i] = np.clip(x + sigma * n, np.float32(0.0), np.float32(1.0))
    return out
