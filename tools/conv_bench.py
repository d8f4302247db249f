"""Micro-benchmark of representative ResNet-50 conv shapes through fav_op_conv2d.  (FAV_* knobs - tile forcing, phase clocks - exist in the
experiments build only: make -C failure_aware_vision_amd/csrc EXPERIMENTS=1.)"""
import ctypes as C, os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import _lib
lib = _lib.load()
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=1920); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="")
a = ap.parse_args()
SHAPES = [  # name, H, Cin, Cout, k, stride, res, drop
    ("L1c1 1x1 256->64", 56, 256, 64, 1, 1, 0, 0), ("L1c2 3x3 64->64", 56, 64, 64, 3, 1, 0, 0),
    ("L1c3 1x1 64->256 +res+drop", 56, 64, 256, 1, 1, 1, 1),
    ("L2c1 1x1 512->128", 28, 512, 128, 1, 1, 0, 0), ("L2c2 3x3 128->128", 28, 128, 128, 3, 1, 0, 0),
    ("L2c3 1x1 128->512 +res+drop", 28, 128, 512, 1, 1, 1, 1),
    ("L3c1 1x1 1024->256", 14, 1024, 256, 1, 1, 0, 0), ("L3c2 3x3 256->256", 14, 256, 256, 3, 1, 0, 0),
    ("L3c3 1x1 256->1024 +res+drop", 14, 256, 1024, 1, 1, 1, 1),
    ("L4c1 1x1 2048->512", 7, 2048, 512, 1, 1, 0, 0), ("L4c2 3x3 512->512", 7, 512, 512, 3, 1, 0, 0),
    ("L4c3 1x1 512->2048 +res+drop", 7, 512, 2048, 1, 1, 1, 1),
    ("T2c1 1x1 256->128 @56", 56, 256, 128, 1, 1, 0, 0), ("T2ds 1x1s2 256->512 @56", 56, 256, 512, 1, 2, 0, 0),
    ("T2c2 3x3s2 128->128 @56", 56, 128, 128, 3, 2, 0, 0),
    ("T3c1 1x1 512->256 @28", 28, 512, 256, 1, 1, 0, 0), ("T3ds 1x1s2 512->1024 @28", 28, 512, 1024, 1, 2, 0, 0),
    ("T3c2 3x3s2 256->256 @28", 28, 256, 256, 3, 2, 0, 0),
    ("X L3c3 plain", 14, 256, 1024, 1, 1, 0, 0), ("X L3c3 +res", 14, 256, 1024, 1, 1, 1, 0), ("X L3c3 +drop", 14, 256, 1024, 1, 1, 0, 1),
    ("X L1c3 plain", 56, 64, 256, 1, 1, 0, 0), ("X L1c3 +res", 56, 64, 256, 1, 1, 1, 0), ("X L1c3 +drop", 56, 64, 256, 1, 1, 0, 1),
    # ViT-B/16 linear layers: 197 rows per frame (H = 197, W = 1), --frames = frames per call
    ("V qkv 768->2304", 197, 768, 2304, 1, 1, 0, 0), ("V proj 768->768 +res", 197, 768, 768, 1, 1, 1, 0),
    ("V fc1 768->3072 gelu", 197, 768, 3072, 1, 1, 0, 0), ("V fc2 3072->768 +res", 197, 3072, 768, 1, 1, 1, 0),
]
tot = 0.0
for name, H, cin, cout, k, stride, res, drop in SHAPES:
    if a.only and a.only not in name: continue
    n = a.frames * (4 if H == 7 else 1)
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    vit = name.startswith("V ")
    W, Wo = (1, 1) if vit else (H, Ho)
    x = (torch.randn(n, H, W, cin, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda") * (2.0 / (k * k * cin)) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda") * 0.1
    r = (torch.randn(n, Ho, Wo, cout, device="cuda")).to(torch.bfloat16) if res else None
    y = torch.empty(n, Ho, Wo, cout, device="cuda", dtype=torch.bfloat16)
    dd = _lib.FavDropoutDesc(3 if drop else -1, 26, 1.0 / (1 - 26 / 256), 4, 0, 256, 0)
    d = _lib.FavConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(),
                         n, H, W, cin, cout, k, k, stride, pad, (2 if "gelu" in name else 0) if vit else 1, 0, 0, dd)
    for _ in range(2): _lib.check(lib.fav_op_conv2d(C.byref(d), None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters): lib.fav_op_conv2d(C.byref(d), None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    M = n * Ho * Wo
    fl = 2.0 * M * cout * k * k * cin
    by = 2.0 * (n * H * W * cin + M * cout * (2 if res else 1) + cout * k * k * cin)
    # scale to the headline workload: 7680 virtual frames
    scale = 7680.0 / n
    tot += ms * scale
    print(f"{name:32s} M={M:8d} {ms:8.3f} ms  {fl/ms/1e9:8.1f} TF/s  {by/ms/1e6:8.0f} GB/s   (x{scale:.0f} -> {ms*scale:6.2f} ms/step)", flush=True)
print(f"sum scaled to 7680 frames: {tot:.2f} ms")
