"""GPU diagnostic: how far is the bf16-MFMA accumulator from the exact sum, compared
with a plain fp32 CPU sum?  Also dumps end-to-end logit error numbers."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from failure_aware_vision_amd import _lib, Backend, synth, weights
from oracle import fav_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ops import run_conv

lib = _lib.load()
rng = np.random.default_rng(0)
out = {}
for (cin, k) in [(64, 1), (64, 3), (256, 3), (512, 3), (2048, 1)]:
    n, H, W, cout = 2, 14, 14, 128
    x = O.bf16_round(np.maximum(rng.standard_normal((n, H, W, cin)), 0).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = np.zeros(cout, np.float32)
    L = O.ConvLayer(cout, cin, k, k, 1, k // 2, w, b)
    cols, ho, wo = O._im2col(x, k, k, 1, k // 2)
    exact = (cols.astype(np.float64) @ w.reshape(cout, -1).T.astype(np.float64)).reshape(n, ho, wo, cout)
    cpu32 = O.conv_acc(x, L)
    res = {}
    for mode in (0, 1):
        g = run_conv(lib, x, w, b, None, 1, k // 2, relu=0, out_f32=1, math_mode=mode)
        res[mode] = g
    scale = np.abs(exact).mean()
    e = lambda a: float(np.abs(a - exact).mean() / scale)
    d = dict(K=k * k * cin, cpu_fp32_err=e(cpu32), mfma_bf16_err=e(res[0]), mfma_f32_err=e(res[1]),
             bf16_round_mismatch_vs_exact_cpu=float((O.bf16_round(cpu32) != O.bf16_round(exact.astype(np.float32))).mean()),
             bf16_round_mismatch_vs_exact_mfma=float((O.bf16_round(res[0]) != O.bf16_round(exact.astype(np.float32))).mean()),
             bf16_round_mismatch_vs_exact_f32mfma=float((O.bf16_round(res[1]) != O.bf16_round(exact.astype(np.float32))).mean()))
    print(d, flush=True)
    out[f"cin{cin}_k{k}"] = d

blob, _ = weights.make_synthetic("resnet50", seed=1)
model = O.parse_blob(blob)
frames = synth.synthetic_frames_u8(8, 224, 224, seed=7)
ol, oc, olg, opb = O.classify(model, frames, O.ClassifyConfig(), return_logits=True)
for mm in ("bf16", "f32_exact"):
    be = Backend("resnet50", blob, max_batch=8, math_mode=mm)
    l, c = be.classify(torch.from_numpy(frames).cuda())
    g = be.logits().cpu().numpy()
    rms = float(np.sqrt(((g - olg) ** 2).mean()) / olg.std())
    srt = np.sort(opb, axis=1); gap = srt[:, -1] - srt[:, -2]
    print(mm, "e2e rms", rms, "labels", l.cpu().numpy().tolist(), ol.tolist(), "gap", np.round(gap, 3).tolist(),
          "dconf", float(np.abs(c.cpu().numpy() - oc).max()), flush=True)
    out["e2e_" + mm] = rms
    be.close()
json.dump(out, open("gpurun_out/diag_precision.json", "w"), indent=1)
