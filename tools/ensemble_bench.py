"""BASELINE configs[3]: 5-member ResNet-50 deep ensemble, the per-GPU share (batch 256 / 8 GPUs = 32 frames, and 256)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import Backend, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--members", type=int, default=5)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
blobs = [weights.make_synthetic("resnet50", seed=1 + m)[0] for m in range(a.members)]
for batch in (32, 256):
    be = Backend("resnet50", blobs, max_batch=batch)
    frames = torch.from_numpy(synth.gaussian_noise_f32(synth.synthetic_frames_u8(batch, 224, 224, seed=21), 3, seed=3)).cuda()
    for _ in range(3):
        be.classify(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        labels, conf = be.classify(frames)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fps = batch * a.steps / dt
    print(json.dumps({"workload": "%d-member ResNet-50 ensemble, 224x224, %d frames per call, 1 MI355X" % (a.members, batch),
                      "frames_per_s": fps, "ms_per_call": 1e3 * dt / a.steps,
                      "tflops_algorithmic": fps * 8.178 * a.members / 1e3}), flush=True)
    be.close()
