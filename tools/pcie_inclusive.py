"""The headline step with the frames handed over in HOST memory (fav_classify_host: one H2D copy on the handle's stream, the
classification, the results back): the PCIe-inclusive rate DESIGN.md section 6 quotes next to `value` (never `value` itself)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from failure_aware_vision_amd import Backend, synth, weights

blob, _ = weights.make_synthetic("resnet50", seed=1)
be = Backend("resnet50", blob, max_batch=256, n_samples=30, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
u8 = synth.synthetic_frames_u8(256, 224, 224, seed=21)
f32 = synth.gaussian_noise_f32(u8, 3, seed=3)
out = {}
for name, host in (("fp32 frames, pageable numpy", f32), ("uint8 frames, pageable numpy", u8),
                   ("fp32 frames, pinned", torch.from_numpy(f32).pin_memory().numpy()), ("resident in HBM (the bench line)", torch.from_numpy(f32).cuda())):
    for _ in range(2):
        be.classify(host)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 8
    for _ in range(steps):
        labels, conf = be.classify(host)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[name] = {"ms_per_step": round(1e3 * dt, 2), "frames_per_s": round(256 / dt, 1)}
    print(name, out[name], flush=True)
print(json.dumps(out))
