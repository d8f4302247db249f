"""On-device corruption generator, the host mirror of the reference's
``VisionSimulator`` controls (platform/backend/vision_simulator.py:12-60): ``set_mode``
(normal / frozen / blank / corrupted), ``set_noise``, ``set_brightness``, ``get_vision_status``,
plus ``apply(frames)`` which actually produces the corrupted frames on the GPU
(``fav_op_corrupt``) — what the browser canvas does in the reference (app.js:782-857) — and
``gaussian(frames, severity)`` for ImageNet-C style noise.  Deterministic in (seed, frame index).
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .synth import GAUSSIAN_NOISE_SIGMA

VALID_MODES = ("normal", "frozen", "blank", "corrupted")
_STATUS = {"normal": "VISION_OK", "frozen": "VISION_FROZEN", "blank": "VISION_BLANK", "corrupted": "VISION_CORRUPTED"}


class Corruptor:
    def __init__(self, seed: int = 0, device: int | None = None):
        import torch
        self._torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("Corruptor needs a gfx950 GPU; there is no CPU fallback")
        self.lib.fav_op_corrupt.restype = C.c_int
        self.lib.fav_op_corrupt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                            C.c_float, C.c_float, C.c_uint64, C.c_int64, C.c_void_p]
        self.seed = int(seed)
        self.reset()

    def reset(self):
        self.mode, self.noise_level, self.brightness = "normal", 0.0, 0.5
        self._last = None          # last frame shown (what "frozen" repeats)
        self._frame_index = 0

    def set_mode(self, mode: str):
        if mode in VALID_MODES:
            self.mode = mode

    def set_noise(self, level: float):
        self.noise_level = max(0.0, min(1.0, float(level)))

    def set_brightness(self, level: float):
        self.brightness = max(0.0, min(1.0, float(level)))

    def get_vision_status(self) -> str:
        return _STATUS[self.mode]

    def _run(self, frames, mode, out_dtype, sigma=0.0, first_index=None):
        torch = self._torch
        frames = frames.contiguous()
        n, H, W, _ = frames.shape
        out = torch.empty(frames.shape, dtype=out_dtype, device=frames.device)
        idx = self._frame_index if first_index is None else int(first_index)
        stream = torch.cuda.current_stream(frames.device).cuda_stream
        _lib.check(self.lib.fav_op_corrupt(frames.data_ptr(), out.data_ptr(), n, H, W, mode, self.noise_level,
                                           self.brightness / 0.5, float(sigma), self.seed, idx, stream))
        return out

    def apply(self, frames):
        """uint8 CUDA frames [n, H, W, 3] of a stream -> the frames the current mode shows."""
        torch = self._torch
        n = int(frames.shape[0])
        if self.mode == "frozen" and self._last is not None:
            out = self._last.unsqueeze(0).expand(n, -1, -1, -1).contiguous()
        else:
            code = {"normal": 0, "frozen": 0, "blank": 1, "corrupted": 2}[self.mode]
            out = self._run(frames, code, torch.uint8)
        self._last = out[-1].clone()
        self._frame_index += n
        return out

    def gaussian(self, frames, severity: int, first_index: int = 0):
        """uint8 frames -> fp32 [0,1] frames with Gaussian noise of ImageNet-C severity 1..5."""
        return self._run(frames, 3, self._torch.float32, GAUSSIAN_NOISE_SIGMA[severity - 1], first_index)
