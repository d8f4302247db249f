"""The stated tolerance against a PURE fp32 PyTorch model (north_star: "outputs match the reference's own PyTorch-CPU
path"; the reference lists torch at requirements.txt:1-2 and never imports it, so the model is this repo's).

The synthetic checkpoint IS an fp32 state_dict (weights.make_synthetic_state_dict; the FAVW blob is its fold), so
tests/torch_models.ResNet("resnet50") loads it as it is: BatchNorm un-folded, fp32 weights, fp32 activations, no bf16
anywhere, eval().  Labels / confidences / top-2 gaps of the 10,000 corrupted frames are stored; the GPU test
prints the agreement of the production mode with them and asserts a bound derived from it.

  python tests/golden/make_fp32_module_fixture.py [n]  # ~10 min on 8 cores -> tests/golden/r50_fp32_module_10k.npz
"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from failure_aware_vision_amd import synth, weights
from oracle.torch_fp32 import ResNet, load_synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
FRAME_SEED, NOISE_SEED, SEVERITY = 21, 3, 3
sd, meta = weights.make_synthetic_state_dict("resnet50", seed=1)
blob, info = weights.from_state_dict("resnet50", sd, bn_eps=meta["bn_eps"])
net, meta = load_synthetic("resnet50", seed=1)
mean = np.asarray(meta["mean"], np.float32); std = np.asarray(meta["std"], np.float32)
n, bs = (int(sys.argv[1]) if len(sys.argv) > 1 else 10000), 50
labels = np.zeros(n, np.int16); conf = np.zeros(n, np.float32); gap = np.zeros(n, np.float32); top2 = np.zeros(n, np.int16)
t0 = time.time()
with torch.no_grad():
    for s in range(0, n, bs):
        u8 = synth.synthetic_frames_u8(bs, 224, 224, seed=FRAME_SEED, start_id=s)
        x = synth.gaussian_noise_f32(u8, SEVERITY, seed=NOISE_SEED, start_id=s)
        xn = (x - mean) / std                                         # plain fp32 normalisation, as torchvision transforms do
        pb = torch.softmax(net(torch.from_numpy(xn.transpose(0, 3, 1, 2).copy())), dim=1).numpy()
        srt = np.argsort(pb, axis=1)
        labels[s:s + bs] = srt[:, -1]; top2[s:s + bs] = srt[:, -2]
        conf[s:s + bs] = pb[np.arange(bs), srt[:, -1]]
        gap[s:s + bs] = pb[np.arange(bs), srt[:, -1]] - pb[np.arange(bs), srt[:, -2]]
        print(s + bs, round(time.time() - t0, 1), flush=True)
np.savez_compressed(os.path.join(HERE, "r50_fp32_module_%dk.npz" % (n // 1000)), labels=labels, conf=conf, gap=gap, second=top2,
                    blob_sha256=info["sha256"],
                    meta="resnet50 seed1 as an fp32 torch nn.Module (BatchNorm un-folded, eval, torch %s CPU); frames seed 21 ids "
                         "0..%d + gaussian noise sev3 seed 3; single pass; gap = top-1 minus top-2 probability" % (torch.__version__, n - 1))
print("done")
