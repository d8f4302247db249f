"""Host side of the drop-in: ``Backend.classify(images) -> (labels, confidences)``.

Mirrors the reference's scorer interface so it plugs into the same seam:

* construction per connection / ``reset()`` / drop on disconnect —
  platform/backend/main.py:110-118, :284-291, :310-317;
* ``analyze_frame(frame) -> {'anomaly_score', 'vision_status', 'metrics'}`` —
  the shape of SignalAnalyzer.analyze_frame, platform/backend/signal_analyzer.py:47-143,
  whose result is fed to ``TrustEngine.update(vision_status, anomaly_score, dt)``
  (main.py:160-168);
* errors at the seam: a malformed frame raises (a caller bug); a failure of the GPU
  path is reported as a NON-OK ``vision_status`` with a numeric ``anomaly_score``
  (main.py:169 rounds the score, and an unseen frame must not count as healthy:
  trust_engine.py:179-190); ``classify`` itself raises.

All arithmetic happens in the HIP library behind include/fav.h.  torch is used
only for device buffers and the current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, weights as _weights

_ARCH = {"resnet18_cifar": _lib.ARCH_RESNET18_CIFAR, "resnet50": _lib.ARCH_RESNET50, "vit_b16": _lib.ARCH_VIT_B16,
         "vit_tiny": _lib.ARCH_VIT_TINY}
_CONF = {"max_softmax": _lib.CONF_MAX_SOFTMAX, "entropy": _lib.CONF_ENTROPY}
_MATH = {"bf16": _lib.MATH_BF16, "f32_exact": _lib.MATH_F32_EXACT}


class Backend:
    def __init__(self, arch: str = "resnet50", blob: bytes | None = None, *, seed_weights: int = 1, device: int | None = None,
                 in_hw=None, max_batch: int = 256, num_classes: int | None = None,
                 n_samples: int = 1, dropout_policy: str = "none", dropout_p: float = 0.0, seed: int = 0,
                 site_mask: int | None = None, temperature: float = 1.0, conf_kind: str = "max_softmax",
                 tau: float = 0.5, math_mode: str = "bf16", mean=None, std=None,
                 chunk_a: int = 0, chunk_b: int = 0, regroup_block: int = -1,
                 tail_min_rows: int = 0, ens_grouped_max: int = 0, vit_streams: int = 0, stem_fused: int = 0):
        import torch
        self._torch = torch
        self._h = None
        self._rules = None
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("failure_aware_vision_amd.Backend needs a gfx950 GPU (torch.cuda.is_available() is "
                               "False); the path has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.arch = arch
        cfg = _lib.FavConfig()
        self.lib.fav_default_config(C.byref(cfg), _ARCH[arch])
        cfg.device = self.device
        if in_hw is not None:
            cfg.in_h, cfg.in_w = int(in_hw[0]), int(in_hw[1])
        if num_classes is not None:
            cfg.num_classes = int(num_classes)
        cfg.max_batch = int(max_batch)
        if mean is not None:
            cfg.mean[:] = [float(m) for m in mean]
        if std is not None:
            cfg.stdev[:] = [float(s) for s in std]
        cfg.n_samples = int(n_samples)
        cfg.site_mask = int(site_mask) if site_mask is not None else _weights.site_mask_for(_ARCH[arch], dropout_policy)
        cfg.dropout_p = float(dropout_p)
        cfg.seed = int(seed)
        cfg.temperature = float(temperature)
        cfg.conf_kind = _CONF[conf_kind]
        cfg.tau = float(tau)
        cfg.math_mode = _MATH[math_mode]
        cfg.chunk_a, cfg.chunk_b, cfg.regroup_block = int(chunk_a), int(chunk_b), int(regroup_block)
        # schedule choices (fav_config, ABI 2): 0 = the build's measured default; results never depend on them
        cfg.tail_min_rows, cfg.ens_grouped_max = int(tail_min_rows), int(ens_grouped_max)
        cfg.vit_streams, cfg.stem_fused = int(vit_streams), int(stem_fused)
        members = list(blob) if isinstance(blob, (list, tuple)) else None   # deep ensemble: one blob per member
        cfg.n_members = len(members) if members else 1
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(self.lib.fav_create(C.byref(cfg), C.byref(h)))
        self._h = h
        if members:
            for i, b in enumerate(members):
                self.load_weights(b, member=i)
        else:
            if blob is None and _ARCH[arch] in _weights.VIT_CFG:
                blob, self.weights_info = _weights.make_synthetic_vit(arch, seed=seed_weights, num_classes=cfg.num_classes,
                                                                     in_hw=(cfg.in_h, cfg.in_w))
            elif blob is None:
                blob, self.weights_info = _weights.make_synthetic(arch, seed=seed_weights, num_classes=cfg.num_classes)
            self.load_weights(blob)
        self.mc = cfg.site_mask != 0 and round(cfg.dropout_p * 256) > 0
        self.T = cfg.n_samples if self.mc else max(1, cfg.n_members)
        self._rules = None   # SignalAnalyzerHIP, created on the first analyze_frame

    # -- lifecycle ------------------------------------------------------------
    def load_weights(self, blob: bytes, member: int = 0):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        _lib.check(self.lib.fav_load_member_weights(self._h, int(member), buf, len(blob)), self._h)

    def reset(self):
        """Scorer reset on mode switch (main.py:222,227,242,288).  The classifier is stateless; the rule
        scorer's previous-gray / frozen-run state is cleared like SignalAnalyzer.reset (signal_analyzer.py:37-39)."""
        if self._rules is not None:
            self._rules.reset()

    def close(self):
        if self._h is not None:
            self.lib.fav_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the hot path ------------------------------------------------------------
    def _layout_of(self, images) -> int:
        torch = self._torch
        dt = images.dtype
        if dt in (torch.uint8, np.uint8):
            return _lib.LAYOUT_NHWC_U8
        if dt in (torch.float32, np.float32):
            return _lib.LAYOUT_NHWC_F32
        raise TypeError(f"frames must be uint8 or float32 NHWC, got {dt}")

    def _check_shape(self, images):
        if images.ndim != 4 or images.shape[1] != self.cfg.in_h or images.shape[2] != self.cfg.in_w or images.shape[3] != 3:
            raise ValueError(f"expected frames of shape (n, {self.cfg.in_h}, {self.cfg.in_w}, 3), got {tuple(images.shape)}")

    def classify_detect(self, images, first_index: int = 0):
        """-> (labels int32[n], confidences fp32[n], fail uint8[n], anomaly_score fp32[n]).
        torch CUDA tensors in -> torch CUDA tensors out (asynchronous on the current
        stream); numpy in -> numpy out (synchronous)."""
        torch = self._torch
        self._check_shape(images)
        layout = self._layout_of(images)
        n = int(images.shape[0])
        if isinstance(images, np.ndarray):
            img = np.ascontiguousarray(images)
            labels = np.empty(n, np.int32); conf = np.empty(n, np.float32)
            fail = np.empty(n, np.uint8); score = np.empty(n, np.float32)
            _lib.check(self.lib.fav_classify_host(self._h, img.ctypes.data, n, layout, int(first_index),
                                                  labels.ctypes.data, conf.ctypes.data, fail.ctypes.data,
                                                  score.ctypes.data), self._h)
            return labels, conf, fail, score
        if not images.is_cuda or images.device.index != self.device:
            raise ValueError(f"frames must live on cuda:{self.device}")
        img = images.contiguous()
        dev = img.device
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        conf = torch.empty(n, dtype=torch.float32, device=dev)
        fail = torch.empty(n, dtype=torch.uint8, device=dev)
        score = torch.empty(n, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.lib.fav_classify_ex(self._h, img.data_ptr(), n, layout, int(first_index), labels.data_ptr(),
                                            conf.data_ptr(), fail.data_ptr(), score.data_ptr(), stream), self._h)
        return labels, conf, fail, score

    def classify_records(self, images, first_index: int = 0, out=None):
        """frames (CUDA tensor) -> int32[n, 2] CUDA tensor of packed (label, confidence bits) records, written by the
        confidence head itself (fav_classify_records).  ``out``: a contiguous int32[n, 2] view to write into - e.g. this
        rank's slot of an all-gather send buffer (distributed.classify_sharded), so nothing is packed or copied."""
        torch = self._torch
        self._check_shape(images)
        layout = self._layout_of(images)
        n = int(images.shape[0])
        if isinstance(images, np.ndarray) or not images.is_cuda or images.device.index != self.device:
            raise ValueError(f"classify_records takes frames on cuda:{self.device}")
        img = images.contiguous()
        if out is None:
            out = torch.empty((n, 2), dtype=torch.int32, device=img.device)
        if out.dtype != torch.int32 or tuple(out.shape) != (n, 2) or not out.is_contiguous() or out.device != img.device:
            raise ValueError("out must be a contiguous int32[n, 2] tensor on the frames' device")
        stream = torch.cuda.current_stream(img.device).cuda_stream
        _lib.check(self.lib.fav_classify_records(self._h, img.data_ptr(), n, layout, int(first_index), out.data_ptr(),
                                                 None, None, stream), self._h)
        return out

    def classify(self, images, first_index: int = 0):
        """The drop-in: frames -> (labels, confidences)."""
        labels, conf, _, _ = self.classify_detect(images, first_index)
        return labels, conf

    def logits(self):
        """fp32 [T, n, num_classes] logits of the last classify call (torch CUDA tensor)."""
        torch = self._torch
        t, n = C.c_int32(), C.c_int32()
        _lib.check(self.lib.fav_get_logits(self._h, None, C.byref(t), C.byref(n), None), self._h)
        out = torch.empty((t.value, n.value, self.cfg.num_classes), dtype=torch.float32, device=f"cuda:{self.device}")
        stream = torch.cuda.current_stream(out.device).cuda_stream
        _lib.check(self.lib.fav_get_logits(self._h, out.data_ptr(), None, None, stream), self._h)
        return out

    # -- profiling (bench.py roofline leg) ----------------------------------------
    def set_profiling(self, enable: bool):
        _lib.check(self.lib.fav_set_profiling(self._h, 1 if enable else 0), self._h)

    def get_profile(self, reset: bool = True) -> dict:
        p = _lib.FavProfile()
        _lib.check(self.lib.fav_get_profile(self._h, C.byref(p), 1 if reset else 0), self._h)
        return {name: dict(ms=p.ms[i], flops=p.flops[i], bytes=p.bytes[i], launches=p.launches[i])
                for i, name in enumerate(_lib.KERNEL_CLASS_NAMES)}

    def get_op_profile(self) -> list:
        """Per-op rows of the static schedule (call after get_profile)."""
        n = C.c_int32()
        _lib.check(self.lib.fav_get_op_profile(self._h, None, 0, C.byref(n)), self._h)
        arr = (_lib.FavOpProfile * n.value)()
        _lib.check(self.lib.fav_get_op_profile(self._h, arr, n.value, C.byref(n)), self._h)
        return [{f: getattr(r, f) for f, _ in _lib.FavOpProfile._fields_ if f != "reserved"} for r in arr]

    # -- the reference seam ----------------------------------------------------------
    #: what the seam reports when the GPU path itself fails (never 'VISION_OK': the trust engine must not recover
    #: reliability on a frame nobody looked at, trust_engine.py:179-190)
    FAILED_STATUS = "VISION_CORRUPTED"

    def analyze_frame(self, frame: np.ndarray, status_provider=None) -> dict:
        """ONE call at the seam (main.py:160): a uint8 HxWx3 frame -> the dict SignalAnalyzer.analyze_frame returns
        (signal_analyzer.py:128-143), with the SAME keys the caller reads (main.py:163-177: ``anomaly_score``,
        ``vision_status``, ``metrics['blur']``, ``metrics['brightness']``, ...).  ``vision_status`` and the
        ``metrics`` entries of the reference are the rule scorer's (computed on the GPU by the fused
        signal-statistics kernel, signal.py); ``anomaly_score`` is the classifier's clamp(1 - confidence), rounded
        to 6 places, always a number (main.py:169 rounds it); ``metrics['classifier']`` = {label, confidence, fail,
        samples} and ``metrics['rule_anomaly_score']`` carry the rest.

        The frame is uploaded once; both kernels are queued on one stream and the host synchronises once.
        ``status_provider(frame) -> str`` replaces the built-in rules (the reference's metric keys are then 0.0).

        Errors.  A frame of the wrong type, dtype or shape is a caller bug: TypeError / ValueError, as
        ``classify`` raises.  A failure of the GPU path (FavError, a HIP error surfacing through torch) does NOT
        read as a healthy frame: the result is ``vision_status = FAILED_STATUS``, ``anomaly_score = 1.0`` and
        ``metrics['error']`` - the engine then decays trust at its corrupted-frame rate (trust_engine.py:218-224) -
        and the rule scorer's previous-frame state is left as it was before the call."""
        torch = self._torch
        if not isinstance(frame, np.ndarray):
            raise TypeError(f"analyze_frame takes a numpy uint8 HxWx3 frame (video_source.py:144-148), got {type(frame).__name__}")
        if frame.dtype != np.uint8:
            raise TypeError(f"analyze_frame takes uint8 pixels, got {frame.dtype}")
        if frame.shape != (self.cfg.in_h, self.cfg.in_w, 3):
            raise ValueError(f"expected a frame of shape ({self.cfg.in_h}, {self.cfg.in_w}, 3), got {tuple(frame.shape)}")
        fr = np.ascontiguousarray(frame)
        ref_metrics = {"blur": 0.0, "brightness": 0.0, "freeze": 0.0, "entropy": 0.0, "raw": {}}
        rule_score = None
        rules, saved = None, None
        # the caller's callback runs OUTSIDE the guarded region: an exception in it is the caller's bug and propagates; only the
        # device path below fails closed
        status = status_provider(frame) if status_provider is not None else None
        try:
            if status_provider is not None:
                labels, conf, fail, score = self.classify_detect(fr[None])
                l0, c0, f0, s0 = int(labels[0]), float(conf[0]), bool(fail[0]), float(score[0])
            else:
                if self._rules is None:
                    from .signal import SignalAnalyzerHIP
                    self._rules = SignalAnalyzerHIP(self.device)
                rules, saved = self._rules, self._rules.save_state()
                dev = torch.from_numpy(fr[None]).to(f"cuda:{self.device}")
                stats_dev = rules.launch_stats(dev)
                labels, conf, fail, score = self.classify_detect(dev)
                packed = torch.stack([labels.to(torch.float32), conf, fail.to(torch.float32), score])   # [4, 1]
                host = torch.cat([stats_dev, packed.view(torch.uint8).flatten()]).cpu().numpy()          # the one sync
                nstat = stats_dev.numel()
                rule = rules.score_stats(rules.parse_stats(host[:nstat].tobytes(), 1))[0]
                status, ref_metrics, rule_score = rule["vision_status"], dict(rule["metrics"]), rule["anomaly_score"]
                vals = host[nstat:].view(np.float32)
                l0, c0, f0, s0 = int(vals[0]), float(vals[1]), bool(vals[2]), float(vals[3])
        except (_lib.FavError, RuntimeError) as e:       # the device path failed: fail closed (see the docstring)
            if rules is not None:
                rules.restore_state(saved)
            return {"anomaly_score": 1.0, "vision_status": self.FAILED_STATUS,
                    "metrics": dict(ref_metrics, error=f"{type(e).__name__}: {e}")}
        metrics = dict(ref_metrics)
        metrics["classifier"] = {"label": l0, "confidence": round(c0, 4), "fail": f0, "samples": self.T}
        if rule_score is not None:
            metrics["rule_anomaly_score"] = rule_score
        return {"anomaly_score": round(s0, 6), "vision_status": status, "metrics": metrics}


def anomaly_score_from_confidence(conf):
    """score = clamp(1 - conf, 0, 1): same [0,1] range contract as signal_analyzer.py:121."""
    return np.clip(1.0 - np.asarray(conf, np.float32), 0.0, 1.0).astype(np.float32)
