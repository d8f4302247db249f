"""Checksum of a few large convolutions through fav_op_conv2d (fixed seeds): for A/B runs of kernel variants selected by
environment variables - the hashes of two runs must agree bit for bit."""
import ctypes as C, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import _lib
lib = _lib.load()
for name, H, cin, cout, k, stride, res, n in [("3x3 256->256", 14, 256, 256, 3, 1, 0, 1000), ("3x3 512->512", 7, 512, 512, 3, 1, 0, 3000),
                                              ("1x1 1024->256", 14, 1024, 256, 1, 1, 0, 1001), ("1x1 512->2048 +res+drop", 7, 512, 2048, 1, 1, 1, 900),
                                              ("3x3s2 256->256", 28, 256, 256, 3, 2, 0, 700)]:
    g = torch.Generator(device="cuda").manual_seed(cin + cout + k)
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    x = (torch.randn(n, H, H, cin, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(cout, k, k, cin, device="cuda", generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    r = torch.randn(n, Ho, Ho, cout, device="cuda", generator=g).to(torch.bfloat16) if res else None
    y = torch.zeros(n, Ho, Ho, cout, device="cuda", dtype=torch.bfloat16)
    dd = _lib.FavDropoutDesc(3 if res else -1, 26, 1.0 / (1 - 26 / 256), 4, 0, 256, 0)
    d = _lib.FavConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(),
                         n, H, H, cin, cout, k, k, stride, pad, 1, 0, 0, dd)
    _lib.check(lib.fav_op_conv2d(C.byref(d), None))
    torch.cuda.synchronize()
    print(f"{name:28s} M={n * Ho * Ho:8d} sha256 {hashlib.sha256(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]}")
