"""Per-op timing table of the static schedule (HIP events around every launch)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import Backend, synth, weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--samples", type=int, default=30)
ap.add_argument("--policy", default="all_blocks")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--chunk-a", type=int, default=0)
ap.add_argument("--chunk-b", type=int, default=0)
ap.add_argument("--regroup-block", type=int, default=-1)
ap.add_argument("--out", default="")
ap.add_argument("--members", type=int, default=1, help="deep ensemble of this many independently seeded members (policy none)")
a = ap.parse_args()
blob = weights.make_synthetic("resnet50", seed=1)[0] if a.members == 1 else [weights.make_synthetic("resnet50", seed=1 + m)[0] for m in range(a.members)]
T = a.samples if a.policy != "none" else 1
be = Backend("resnet50", blob, max_batch=a.batch, n_samples=T, dropout_policy=a.policy,
             dropout_p=0.1 if a.policy != "none" else 0.0, seed=4, chunk_a=a.chunk_a, chunk_b=a.chunk_b,
             regroup_block=a.regroup_block)
frames = torch.from_numpy(synth.synthetic_frames_u8(a.batch, 224, 224, seed=21)).cuda()
be.classify(frames); torch.cuda.synchronize()
be.set_profiling(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    be.classify(frames)
e1.record(); torch.cuda.synchronize()
cls = be.get_profile()
rows = be.get_op_profile()
kinds = ["im2col", "conv", "maxpool", "avgpool", "dropout", "tail", "drop+red", "stem+pool"]
tot = sum(r["ms"] for r in rows)
print(f"wall {e0.elapsed_time(e1)/a.steps:.2f} ms/step, sum of kernel ms {tot/a.steps:.2f}, policy {a.policy} T={T} batch {a.batch}")
print(f"{'op':>3} {'kind':8} {'in':>16} {'out':>14} {'k':>3} {'s':>2} {'launch':>6} {'ms/step':>8} {'%':>5} {'TF/s':>7} {'GB/s':>7}")
for r in rows:
    if r["launches"] == 0:
        continue
    ms = r["ms"] / a.steps
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
    gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
    print(f"{r['op_index']:>3} {kinds[r['kind']]:8} {r['H']:>4}x{r['W']:<4}x{r['Cin']:<5} {r['Ho']:>3}x{r['Wo']:<3}x{r['Cout']:<5} "
          f"{r['kh']:>3} {r['stride']:>2} {r['launches']//a.steps:>6} {ms:>8.3f} {100*r['ms']/tot:>5.1f} {tf:>7.1f} {gb:>7.0f}")
if a.out:
    json.dump(dict(rows=rows, classes=cls, steps=a.steps), open(a.out, "w"))
