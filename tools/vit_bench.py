"""ViT-B/16 (BASELINE configs[4], one GPU's share): frames/s of `classify` at a given batch; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import Backend, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--streams", type=int, default=0, help="fav_config.vit_streams (0 = the library's choice)")
a = ap.parse_args()
blob, _ = weights.make_synthetic_vit("vit_b16", seed=1)
be = Backend("vit_b16", blob, max_batch=a.batch, temperature=1.5, conf_kind="entropy", vit_streams=a.streams)
frames = torch.from_numpy(synth.synthetic_frames_u8(a.batch, 224, 224, seed=21)).cuda()
for _ in range(3):
    be.classify(frames)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(a.steps):
    be.classify(frames)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
gflop = 2 * (196 * 768 * 768 + 12 * (197 * 768 * (3 * 768 + 768 + 2 * 3072) + 2 * 12 * 197 * 197 * 64)) / 1e9
print(f"vit_b16 batch {a.batch} streams {a.streams}: {ms:.3f} ms/call, {a.batch / ms * 1e3:.0f} frames/s, {a.batch * gflop / ms:.0f} TF/s "
      f"({gflop:.1f} GFLOP/frame)")
