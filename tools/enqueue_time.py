"""Host enqueue time against GPU time of one classify call, for the small configurations (is a call launch-bound?).

Per configuration: the GPU is idle when the call starts; `enqueue_ms` = until the call returns (every launch queued),
`total_ms` = until the stream is drained.  A call whose enqueue time is close to its total time is bound by the host's
launch rate, not by the kernels."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from failure_aware_vision_amd import Backend, synth, weights


def run(name, be, frames, steps=20):
    for _ in range(3):
        be.classify(frames)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        be.classify(frames)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    # back to back, as the benchmark runs it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        be.classify(frames)
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "enqueue_ms": round(1e3 * float(np.median(enq)), 3), "total_ms": round(1e3 * float(np.median(tot)), 3),
                      "back_to_back_ms": round(1e3 * b2b, 3)}), flush=True)


def frames_of(n):
    return torch.from_numpy(synth.gaussian_noise_f32(synth.synthetic_frames_u8(n, 224, 224, seed=21), 3, seed=3)).cuda()


which = sys.argv[1:] or ["ens5", "vit", "single", "one"]
if "ens5" in which:
    blobs = [weights.make_synthetic("resnet50", seed=1 + m)[0] for m in range(5)]
    be = Backend("resnet50", blobs, max_batch=32)
    run("ensemble5@32", be, frames_of(32)); be.close()
if "vit" in which:
    be = Backend("vit_b16", weights.make_synthetic_vit("vit_b16", seed=1)[0], max_batch=64, conf_kind="entropy", temperature=1.5)
    run("vit_b16@64", be, frames_of(64)); be.close()
if "single" in which:
    blob = weights.make_synthetic("resnet50", seed=1)[0]
    for n in (32, 256):
        be = Backend("resnet50", blob, max_batch=n)
        run("single_pass@%d" % n, be, frames_of(n)); be.close()
if "one" in which:
    blob = weights.make_synthetic("resnet50", seed=1)[0]
    be = Backend("resnet50", blob, max_batch=1, n_samples=30, dropout_p=0.1, dropout_policy="all_blocks")
    run("mc30@1", be, frames_of(1)); be.close()
