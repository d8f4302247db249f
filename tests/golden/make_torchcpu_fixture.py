"""Independent labels for the production path: oracle/torch_cpu.py (torch.nn.functional fp32 MKL-DNN convolutions in
whatever summation order the library picks, bf16 rounding at every layer boundary as the numerical contract says)
on the same 10,000 corrupted frames as the oracle fixtures, single deterministic pass.  Unlike the MFMA-model oracle this
one is NOT fitted to the GPU's adder, so agreement with it is independent evidence.

  python tests/golden/make_torchcpu_fixture.py        # ~12 min on 8 cores -> tests/golden/r50_torchcpu_10k.npz
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
n, bs = 10000, 50
labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gap = np.zeros(n, np.float32); top2 = np.zeros(n, np.int16)
t0 = time.time()
for s in range(0, n, bs):
    u8 = synth.synthetic_frames_u8(bs, 224, 224, seed=FRAME_SEED, start_id=s)
    x = synth.gaussian_noise_f32(u8, SEVERITY, seed=NOISE_SEED, start_id=s)
    l, c, lg, pb = TC.classify(model, x, O.ClassifyConfig(), return_logits=True, net=net)
    srt = np.argsort(pb, axis=1)
    labels[s:s + bs] = l; conf[s:s + bs] = c
    top2[s:s + bs] = srt[:, -2]
    gap[s:s + bs] = pb[np.arange(bs), srt[:, -1]] - pb[np.arange(bs), srt[:, -2]]
    if (s // bs) % 10 == 9:
        print(s + bs, round(time.time() - t0, 1), flush=True)
np.savez_compressed(os.path.join(HERE, "r50_torchcpu_10k.npz"), labels=labels, conf=conf, gap=gap, second=top2,
                    blob_sha256=info["sha256"],
                    meta="resnet50 seed1; frames seed 21 ids 0..9999 + gaussian noise sev3 seed 3; single pass; oracle/torch_cpu.py "
                         "(torch %s fp32 CPU convolutions, bf16 layer boundaries); gap = top-1 minus top-2 probability" % __import__("torch").__version__)
print("done")
