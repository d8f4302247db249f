"""The ViT encoder's four linear layers at the per-GPU share of BASELINE configs[4] (64 frames: 12 608 rows) and at 512 frames:
chained stream-K GEMM (fav_op_linear_streamk) against the tile-per-block kernel (fav_op_conv2d), same tensors, bit-compared.
With the experiments build (make EXPERIMENTS=1): FAV_CONV_BIG=0 FAV_CONV_BK=32 makes the tile kernel the encoder's 128 x 128 x 32 form, FAV_SK_DBG=1
prints where a stream-K step's ticks go, FAV_SK_NOHANDOFF=1 times it without its hand-offs (wrong results)."""
import ctypes as C, os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from failure_aware_vision_amd import _lib
lib = _lib.load()
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=64); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--rows", type=int, default=0)
a = ap.parse_args()
rows = a.rows or a.frames * 197


def timed(fn, iters):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot_sk = tot_cv = 0.0
for name, K, N, act, res in [("qkv", 768, 2304, 0, 0), ("proj", 768, 768, 0, 1), ("fc1+gelu", 768, 3072, 2, 0), ("fc2", 3072, 768, 0, 1)]:
    x = (torch.randn(rows, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * (1.0 / K) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda") * 0.1
    r = torch.randn(rows, N, device="cuda").to(torch.bfloat16) if res else None
    y1, y2 = torch.empty(rows, N, device="cuda", dtype=torch.bfloat16), torch.empty(rows, N, device="cuda", dtype=torch.bfloat16)
    nd = _lib.FavDropoutDesc(-1, 0, 1.0, 0, 0, 1, 0)
    cd = _lib.FavConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y1.data_ptr(), 1, rows, 1, K, N, 1, 1, 1, 0, act, 0, 0, nd)
    ld = _lib.FavLinearDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y2.data_ptr(), rows, K, N, act)
    t_cv = timed(lambda: lib.fav_op_conv2d(C.byref(cd), None), a.iters)
    t_sk = timed(lambda: _lib.check(lib.fav_op_linear_streamk(C.byref(ld), None)), a.iters)
    fl = 2.0 * rows * K * N
    tot_sk += t_sk; tot_cv += t_cv
    print(f"{name:9s} rows {rows} K {K:4d} N {N:4d}: tile kernel {t_cv:7.1f} us ({fl / t_cv / 1e6:6.0f} TF/s)   stream-K {t_sk:7.1f} us ({fl / t_sk / 1e6:6.0f} TF/s)"
          f"   x{t_cv / t_sk:4.2f}   bit-identical {torch.equal(y1, y2)}", flush=True)
print(f"sum: tile kernel {tot_cv:.1f} us, stream-K {tot_sk:.1f} us")
