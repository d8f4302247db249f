"""Micro-benchmark of the fused bottleneck tail (fav_op_bottleneck_tail) against the separate launches it replaces,
same process, same tensors, results compared bit for bit.  FAV_CONV_DBG=1 adds per-block phase clocks (FAV_CONV_DBG_DUMP=file: the raw
stamps for tools/phase_overlap.py) - like every FAV_* knob that needs the experiments build: make -C failure_aware_vision_amd/csrc EXPERIMENTS=1."""
import ctypes as C, os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import _lib
lib = _lib.load()
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=1920); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", default="")
a = ap.parse_args()
SHAPES = [  # name, H, Cmid, Nred, has3x3
    ("L1 tail 3x3+c+a 64/256->64", 56, 64, 64, 1), ("L1->L2 tail 64/256->128", 56, 64, 128, 1),
    ("L2 tail no3x3 128/512->128", 28, 128, 128, 0), ("L2 tail 3x3+c+a 128/512->128", 28, 128, 128, 1),
    ("L2 last tail 3x3+c 128/512", 28, 128, 0, 1),
    ("L3 conv_c alone 256/1024", 14, 256, 0, 0),
    ("L3 3x3 + conv_c 256/1024", 14, 256, 0, 1),
    ("L4 conv_c alone 512/2048", 7, 512, 0, 0),
    ("L3 conv_c + next reduce 256/1024->256", 14, 256, 256, 0),
    ("probe 64/256 3x3+c at 28x28 (51 KB of LDS: three blocks per CU)", 28, 64, 0, 1),
]


def timed(fn, iters):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, H, cmid, nred, h3 in SHAPES:
    if a.only and a.only not in name: continue
    n, cout = a.frames, 4 * cmid
    bf = lambda *shape, s=0.5: (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
    x = bf(n, H, H, cmid).clamp_(min=0)
    wb, bb = bf(cmid, 3, 3, cmid, s=(2.0 / (9 * cmid)) ** 0.5), torch.randn(cmid, device="cuda") * 0.1
    wc, bc = bf(cout, 1, 1, cmid, s=(1.0 / cmid) ** 0.5), torch.randn(cout, device="cuda") * 0.1
    wa, ba = (bf(nred, 1, 1, cout, s=(2.0 / cout) ** 0.5), torch.randn(nred, device="cuda") * 0.1) if nred else (None, None)
    res = bf(n, H, H, cout).clamp_(min=0)
    y, y2 = torch.empty_like(res), torch.empty_like(res)
    t2 = torch.empty_like(x)
    t1n, t1n2 = (torch.empty(n, H, H, nred, device="cuda", dtype=torch.bfloat16) for _ in range(2)) if nred else (None, None)
    dd = _lib.FavDropoutDesc(3, 26, 1.0 / (1 - 26 / 256), 4, 0, 256, 0)
    nd = _lib.FavDropoutDesc(-1, 0, 1.0, 0, 0, 1, 0)
    if os.environ.get("TAIL_NODROP"): dd = nd      # what the dropout site costs: the same launch without it
    ptr = lambda t: t.data_ptr() if t is not None else None
    td = _lib.FavTailDesc(ptr(x), ptr(wb) if h3 else None, ptr(bb) if h3 else None, ptr(wc), ptr(bc), ptr(res), ptr(y), ptr(wa), ptr(ba),
                          ptr(t1n), n, H, H, cmid, nred, dd)
    c1 = _lib.FavConvDesc(ptr(x), ptr(wb), ptr(bb), None, ptr(t2), n, H, H, cmid, cmid, 3, 3, 1, 1, 1, 0, 0, nd)
    c2 = _lib.FavConvDesc(ptr(t2 if h3 else x), ptr(wc), ptr(bc), ptr(res), ptr(y2), n, H, H, cmid, cout, 1, 1, 1, 0, 1, 0, 0, dd)
    c3 = _lib.FavConvDesc(ptr(y2), ptr(wa), ptr(ba), None, ptr(t1n2), n, H, H, cout, nred, 1, 1, 1, 0, 1, 0, 0, nd) if nred else None

    def fused(): _lib.check(lib.fav_op_bottleneck_tail(C.byref(td), None))

    def separate():
        if h3: _lib.check(lib.fav_op_conv2d(C.byref(c1), None))
        _lib.check(lib.fav_op_conv2d(C.byref(c2), None))
        if nred: _lib.check(lib.fav_op_conv2d(C.byref(c3), None))
    tf, ts = timed(fused, a.iters), timed(separate, a.iters)
    ok = torch.equal(y, y2) and (not nred or torch.equal(t1n, t1n2))
    M = n * H * H
    by = 2.0 * M * (cmid + 2 * cout + nred)
    scale = 7680.0 / n
    print(f"{name:34s} fused {tf:7.3f} ms ({by / tf / 1e6:6.0f} GB/s)  separate {ts:7.3f} ms  -> x{ts / tf:4.2f}  bit-identical {ok}"
          f"   (x{scale:.0f}: {tf * scale:6.2f} vs {ts * scale:6.2f} ms/step)", flush=True)
