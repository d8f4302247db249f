"""The reference's live-mode scorer on the GPU.

``SignalAnalyzerHIP`` mirrors ``SignalAnalyzer`` (platform/backend/signal_analyzer.py:18-171):
same constructor-less lifecycle, ``reset()``, ``analyze_frame(frame) -> dict`` with the same
keys and roundings, plus ``analyze_frames(frames)`` for a batch of consecutive frames.  The
four per-pixel statistics come from ONE fused HIP pass per frame (``fav_op_signal_stats``);
the scalar scoring and status rules below are host code, as in the reference.

Parity is unpinned: the reference has no test of this scorer and OpenCV is absent from the
build image, so the pixel arithmetic follows OpenCV's documented 8-bit semantics (restated
in oracle/signal_oracle.py, which the tests compare against).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

# fusion weights and thresholds (signal_analyzer.py:22-34)
W_BLUR, W_BRIGHT, W_FREEZE, W_ENTROPY = 0.35, 0.25, 0.15, 0.25
FREEZE_DIFF, FREEZE_RUN = 1.0, 5
BLANK_LO, BLANK_HI = 15, 245
ENTROPY_LO, ENTROPY_HI = 2.0, 7.5
SHARP_LAPLACIAN = 500.0


class FavSignalStats(C.Structure):
    _fields_ = [("lap_var", C.c_double), ("mean", C.c_double), ("mean_diff", C.c_double), ("entropy", C.c_float),
                ("has_prev", C.c_int32), ("sum_lap", C.c_int64), ("sum_lap2", C.c_int64), ("sum_gray", C.c_uint32),
                ("sum_absdiff", C.c_uint32), ("hist", C.c_uint32 * 256)]


def _clip01(x: float) -> float:
    return max(0.0, min(1.0, x))


def score_frame(lap_var: float, mean: float, mean_diff, entropy: float, frozen_run: int):
    """Scalar part of analyze_frame (signal_analyzer.py:66-143).  mean_diff None = first frame.
    Returns (result dict, new frozen_run)."""
    blur = _clip01(1.0 - lap_var / SHARP_LAPLACIAN)
    bright = _clip01(abs(mean - 128.0) / 128.0)
    if mean_diff is None:
        freeze, shown_diff = 0.0, 10.0                      # the reference's first-frame placeholder
    else:
        frozen_run = frozen_run + 1 if mean_diff < FREEZE_DIFF else 0
        freeze = 1.0 if frozen_run >= FREEZE_RUN else (0.3 * frozen_run / FREEZE_RUN if frozen_run > 0 else 0.0)
        shown_diff = mean_diff
    if entropy < 4.0:
        ent = _clip01((4.0 - entropy) / 4.0)
    elif entropy > 7.0:
        ent = _clip01((entropy - 7.0) / 1.5)
    else:
        ent = 0.0
    score = _clip01(W_BLUR * blur + W_BRIGHT * bright + W_FREEZE * freeze + W_ENTROPY * ent)
    if mean < BLANK_LO or mean > BLANK_HI:                    # priority order of _derive_status (:159-171)
        status = "VISION_BLANK"
    elif frozen_run >= FREEZE_RUN:
        status = "VISION_FROZEN"
    elif entropy < ENTROPY_LO or entropy > ENTROPY_HI:
        status = "VISION_CORRUPTED"
    else:
        status = "VISION_OK"
    return {
        "anomaly_score": round(score, 6),
        "vision_status": status,
        "metrics": {"blur": round(blur, 4), "brightness": round(bright, 4), "freeze": round(freeze, 4),
                    "entropy": round(ent, 4),
                    "raw": {"laplacian_var": round(lap_var, 2), "mean_brightness": round(mean, 1),
                            "frame_diff": round(shown_diff, 2), "entropy": round(entropy, 3)}},
    }, frozen_run


class SignalAnalyzerHIP:
    def __init__(self, device: int | None = None):
        import torch
        self._torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("SignalAnalyzerHIP needs a gfx950 GPU; there is no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.lib.fav_op_signal_stats.restype = C.c_int
        self.lib.fav_op_signal_stats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
        self.reset()

    def reset(self):
        self._prev_gray = None       # device uint8 [H, W]
        self._frame_count = 0
        self._consecutive_frozen = 0

    def save_state(self):
        """Snapshot of the temporal state (previous gray plane, frame count, frozen run), for callers that must be
        able to undo a launch_stats / score_stats pair whose device work failed (Backend.analyze_frame)."""
        return (self._prev_gray, self._frame_count, self._consecutive_frozen)

    def restore_state(self, state):
        self._prev_gray, self._frame_count, self._consecutive_frozen = state

    def launch_stats(self, frames):
        """Queue the fused statistics pass for uint8 [n, H, W, 3] frames (numpy or CUDA tensor) on the current
        stream; returns the device byte tensor holding n fav_signal_stats records (no host sync)."""
        torch = self._torch
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames)).to(f"cuda:{self.device}")
        frames = frames.contiguous()
        n, H, W, ch = frames.shape
        if ch != 3 or frames.dtype != torch.uint8:
            raise ValueError("frames must be uint8 [n, H, W, 3] (BGR)")
        out = torch.empty(n * C.sizeof(FavSignalStats), dtype=torch.uint8, device=frames.device)
        last = torch.empty((H, W), dtype=torch.uint8, device=frames.device)
        prev = self._prev_gray
        if prev is not None and tuple(prev.shape) != (H, W):
            prev = None
        stream = torch.cuda.current_stream(frames.device).cuda_stream
        _lib.check(self.lib.fav_op_signal_stats(frames.data_ptr(), n, H, W, prev.data_ptr() if prev is not None else None,
                                                last.data_ptr(), out.data_ptr(), stream))
        self._prev_gray = last
        return out

    @staticmethod
    def parse_stats(host_bytes: bytes, n: int) -> list:
        return [FavSignalStats.from_buffer_copy(host_bytes, i * C.sizeof(FavSignalStats)) for i in range(n)]

    def stats(self, frames):
        """frames: uint8 [n, H, W, 3] (numpy or CUDA tensor) -> list of FavSignalStats (host)."""
        n = int(frames.shape[0])
        return self.parse_stats(self.launch_stats(frames).cpu().numpy().tobytes(), n)

    def score_stats(self, stats) -> list:
        """The scalar scoring + status rules over a list of FavSignalStats (advances the frozen-run state)."""
        res = []
        for st in stats:
            self._frame_count += 1
            r, self._consecutive_frozen = score_frame(st.lap_var, st.mean, st.mean_diff if st.has_prev else None,
                                                      float(st.entropy), self._consecutive_frozen)
            res.append(r)
        return res

    def analyze_frames(self, frames) -> list:
        return self.score_stats(self.stats(frames))

    def analyze_frame(self, frame: np.ndarray) -> dict:
        return self.analyze_frames(np.ascontiguousarray(frame)[None])[0]
