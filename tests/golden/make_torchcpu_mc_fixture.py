"""Independent labels / confidences for the HEADLINE configuration: oracle/torch_cpu.py (torch.nn.functional fp32 CPU
convolutions, bf16 rounding at every layer boundary, the library's own summation order) on 256 corrupted frames - the
headline batch - with
MC-Dropout T = 30, `all_blocks`, p = 0.1, seed 4 - the same Philox masks, prefix caching and mean-of-softmax head as the
GPU path, none of its arithmetic.  (make_torchcpu_fixture.py is the single-pass counterpart.)

  python tests/golden/make_torchcpu_mc_fixture.py          # ~11 min on 8 cores -> tests/golden/r50_torchcpu_mc30_256.npz
  python tests/golden/make_torchcpu_mc_fixture.py 1000     # ~45 min                -> tests/golden/r50_torchcpu_mc30_1000.npz (resumes)
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from failure_aware_vision_amd import synth, weights
from oracle import fav_oracle as O
from oracle import torch_cpu as TC

HERE = os.path.dirname(os.path.abspath(__file__))
FRAME_SEED, NOISE_SEED, SEVERITY = 21, 3, 3
blob, info = weights.make_synthetic("resnet50", seed=1)
model = O.parse_blob(blob)
net = TC.TorchNet(model)
n, bs = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 4
cfg = O.ClassifyConfig(n_samples=30, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4)
labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gap = np.zeros(n, np.float32); top2 = np.zeros(n, np.int16)
part = os.path.join(HERE, "_torchcpu_mc_partial_%d.npz" % n)
done = 0
if os.path.exists(part):
    dd = np.load(part)
    if str(dd["blob_sha256"]) == info["sha256"]:
        done = int(dd["done"]); labels, conf, gap, top2 = dd["labels"], dd["conf"], dd["gap"], dd["second"]
t0 = time.time()
for s in range(done, n, bs):
    u8 = synth.synthetic_frames_u8(bs, 224, 224, seed=FRAME_SEED, start_id=s)
    x = synth.gaussian_noise_f32(u8, SEVERITY, seed=NOISE_SEED, start_id=s)
    l, c, lg, pb = TC.classify(model, x, cfg, img_ids=np.arange(s, s + bs), return_logits=True, net=net, stack_samples=True)
    srt = np.argsort(pb, axis=1)
    labels[s:s + bs] = l; conf[s:s + bs] = c
    top2[s:s + bs] = srt[:, -2]
    gap[s:s + bs] = pb[np.arange(bs), srt[:, -1]] - pb[np.arange(bs), srt[:, -2]]
    print(s + bs, round(time.time() - t0, 1), flush=True)
    if (s // bs) % 8 == 7:
        np.savez(part, done=s + bs, labels=labels, conf=conf, gap=gap, second=top2, blob_sha256=info["sha256"])
np.savez_compressed(os.path.join(HERE, "r50_torchcpu_mc30_%d.npz" % n), labels=labels, conf=conf, gap=gap, second=top2,
                    blob_sha256=info["sha256"],
                    meta="resnet50 seed1; frames seed 21 ids 0..%d + gaussian noise sev3 seed 3; MC-Dropout T=30 all_blocks p=0.1 seed 4; "
                         "oracle/torch_cpu.py (torch %s fp32 CPU convolutions, bf16 layer boundaries); gap = top-1 minus top-2 mean "
                         "probability" % (n - 1, __import__("torch").__version__))
if os.path.exists(part):
    os.remove(part)
print("done")
