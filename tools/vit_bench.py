"""Throughput of the ViT-B/16 path (BASELINE configs[4] shape: 224x224, 64 frames per GPU = batch 512 over 8 GPUs)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import Backend, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
a = ap.parse_args()
blob, info = weights.make_synthetic_vit("vit_b16", seed=1)
be = Backend("vit_b16", blob, max_batch=a.batch, temperature=1.5, conf_kind="entropy")
frames = torch.from_numpy(synth.gaussian_noise_f32(synth.synthetic_frames_u8(a.batch, 224, 224, seed=21), 3, seed=3)).cuda()
for _ in range(a.warmup):
    be.classify(frames)
torch.cuda.synchronize()
be.set_profiling(True)
t0 = time.perf_counter()
for _ in range(a.steps):
    labels, conf = be.classify(frames)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = be.get_profile()
gflop = 35.1   # per frame, SURVEY.md section 8d config 5
fps = a.batch * a.steps / dt
print(json.dumps({"workload": "ViT-B/16 224x224 single pass, entropy confidence (T=1.5), batch %d on 1 MI355X" % a.batch,
                  "frames_per_s": fps, "ms_per_batch": 1e3 * dt / a.steps, "tflops_algorithmic": fps * gflop / 1e3,
                  "kernel_ms_per_batch": {k: v["ms"] / a.steps for k, v in prof.items() if v["ms"] > 0},
                  "labels_distinct": int(len(set(labels.cpu().tolist())))}))
