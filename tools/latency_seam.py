"""Latency of the drop-in call at the reference's own operating point: one 320x240 uint8
frame per call (video_source.py:29-30), ResNet-50, MC-Dropout T=30, from host frame to dict."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from failure_aware_vision_amd import Backend, synth, weights
from failure_aware_vision_amd.signal import SignalAnalyzerHIP
from failure_aware_vision_amd.trust import TrustEngine
blob, _ = weights.make_synthetic("resnet50", seed=1)
for T, pol in ((1, "none"), (30, "all_blocks"), (30, "last_layer")):
    be = Backend("resnet50", blob, in_hw=(240, 320), max_batch=1, n_samples=T, dropout_policy=pol,
                 dropout_p=0.1 if pol != "none" else 0.0, seed=4)
    rules = SignalAnalyzerHIP()
    eng = TrustEngine()
    frames = [synth.synthetic_frame_u8(240, 320, 9, i) for i in range(40)]
    lat = []
    for i, f in enumerate(frames):
        t0 = time.perf_counter()
        status = rules.analyze_frame(f)["vision_status"]
        out = be.analyze_frame(f, status_provider=lambda _f: status)
        eng.update(out["vision_status"], out["anomaly_score"], 1 / 30)
        lat.append((time.perf_counter() - t0) * 1e3)
    print(f"T={T:2d} {pol:10s}: per-frame host-to-dict latency p50 {statistics.median(lat[5:]):.2f} ms, "
          f"max {max(lat[5:]):.2f} ms (30 Hz tick = 33.3 ms)", flush=True)
    be.close()
