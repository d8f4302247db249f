"""CPU restatement of failure_aware_vision_amd/csrc corrupt_kernel (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference's generator is browser JavaScript driven by Math.random
(platform/frontend/js/app.js:789-798,820-822,835-850) and has no reproducible output; this file
restates the build's own definition (same Philox streams, same fp32 operation order)."""
import numpy as np

from .fav_oracle import philox4x32_10

f32 = np.float32


def u01(x):
    return (x >> np.uint32(8)).astype(f32) * f32(1.0 / 16777216.0)


def corrupt(frames, mode, level, gain, sigma, seed, first_index):
    n, H, W, _ = frames.shape
    px = np.arange(H * W, dtype=np.uint32)[None, :]
    fr = (np.arange(n, dtype=np.uint32) + np.uint32(first_index))[:, None]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = frames.reshape(n, H * W, 3).astype(f32)
    if mode == 3:
        a = philox4x32_10(px, fr, np.uint32(3), np.uint32(0), k0, k1)
        r0 = np.sqrt(f32(-2.0) * np.log(f32(1.0) - u01(a[0])), dtype=f32)
        r1 = np.sqrt(f32(-2.0) * np.log(f32(1.0) - u01(a[2])), dtype=f32)
        t0, t1 = f32(6.2831853071795864) * u01(a[1]), f32(6.2831853071795864) * u01(a[3])
        nz = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1)], axis=-1).astype(f32)
        return np.clip(c * f32(1.0 / 255.0) + f32(sigma) * nz, 0, 1).astype(f32).reshape(n, H, W, 3)
    if mode == 1:
        out = np.empty((n, H, W, 3), np.uint8)
        out[...] = (2, 2, 4)
        return out
    u = philox4x32_10(px, fr, np.uint32(mode), np.uint32(0), k0, k1)
    v = c * f32(gain)
    if mode == 0:
        nz = ((u01(u[0]) - f32(0.5)) * f32(255.0)) * f32(level)
        v = v + nz[..., None]
    else:
        hit = u01(u[0]) > f32(0.8)
        g = np.stack([u01(u[1]) * f32(255.0), np.zeros_like(u01(u[1])), u01(u[2]) * f32(255.0)], axis=-1)
        v = np.where(hit[..., None], g, v).astype(f32)
        y = (px // np.uint32(W)).astype(f32)
        for b in range(6):
            q = philox4x32_10(np.uint32(b), fr, np.uint32(7), np.uint32(0), k0, k1)
            by = np.floor(u01(q[0]) * f32(H))
            bh = np.floor(f32(2.0) + u01(q[1]) * f32(12.0))
            al = f32(0.4) + u01(q[2]) * f32(0.5)
            inbar = (y >= by) & (y < by + bh)
            tgt = np.array([255.0, 0.0, 170.0], f32)
            v = np.where(inbar[..., None], v + (tgt - v) * al[..., None], v).astype(f32)
    return np.rint(np.clip(v, 0, 255)).astype(np.uint8).reshape(n, H, W, 3)
