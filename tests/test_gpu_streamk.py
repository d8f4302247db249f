"""The chained stream-K GEMM of the ViT encoder (fav_op_linear_streamk, gemm_streamk_kernel) against the tile-per-block kernel it
replaces (fav_op_conv2d with kh = kw = 1): the K steps are dealt out evenly over a persistent grid and a tile's partial accumulator is
handed from one workgroup to the next, never re-associated, so the two must agree BIT FOR BIT at every shape - and, through the
conv kernel's own tests, with the MFMA-model oracle (no reference counterpart: platform/backend/main.py:160 is the slot)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from failure_aware_vision_amd import _lib  # noqa: E402


def _run(rows, K, N, act, with_res, seed):
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(rows, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * (1.0 / K) ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    res = torch.randn(rows, N, device="cuda", generator=g).to(torch.bfloat16) if with_res else None
    # reference: the tile-per-block kernel, residual added IN PLACE as the encoder does it
    y_ref = res.clone() if with_res else torch.full((rows, N), 7.0, device="cuda", dtype=torch.bfloat16)
    nd = _lib.FavDropoutDesc(-1, 0, 1.0, 0, 0, 1, 0)
    cd = _lib.FavConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y_ref.data_ptr() if with_res else None, y_ref.data_ptr(),
                          1, rows, 1, K, N, 1, 1, 1, 0, act, 0, 0, nd)
    _lib.check(lib.fav_op_conv2d(C.byref(cd), None))
    y = res.clone() if with_res else torch.full((rows, N), -3.0, device="cuda", dtype=torch.bfloat16)
    guard = torch.full((4096,), 1234.0, device="cuda", dtype=torch.bfloat16)   # allocated right behind y on a fresh pool, most of the time
    ld = _lib.FavLinearDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr() if with_res else None, y.data_ptr(), rows, K, N, act)
    for _ in range(2):          # twice: the second launch reuses workspace slots and flags under a new epoch
        if with_res:
            y.copy_(res)
        _lib.check(lib.fav_op_linear_streamk(C.byref(ld), None))
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), \
        f"stream-K differs from the tile kernel: rows {rows} K {K} N {N} act {act} res {with_res}: " \
        f"{int((y.view(torch.int16) != y_ref.view(torch.int16)).sum())} elements"
    assert torch.all(guard == 1234.0)


@pytest.mark.parametrize("rows,K,N,act,with_res", [
    (12608, 768, 2304, 0, False),      # ViT-B/16 at 64 frames: qkv
    (12608, 768, 768, 0, True),        # proj + residual in place (594 tiles: two workgroups per CU)
    (12608, 768, 3072, 2, False),      # fc1 + GELU
    (12608, 3072, 768, 0, True),       # fc2 + residual in place
    (12544, 768, 768, 0, False),       # patch embedding
    (5505, 768, 768, 1, True),         # 44 x 6 = 264 tiles: one workgroup per CU; ragged last row tile (1 valid row); ReLU
    (33000, 128, 256, 0, False),       # K = 4 steps per tile, 516 tiles
    (100864, 768, 768, 0, True),       # 512 frames: 4 728 tiles, three workgroups per CU
    (4929, 3072, 3072, 2, False),      # 39 x 24 = 936 tiles, long K
])
def test_streamk_equals_tile_kernel_bitwise(rows, K, N, act, with_res):
    _run(rows, K, N, act, with_res, seed=rows + K + N)


def test_streamk_rejects_what_it_cannot_balance():
    lib = _lib.load()
    x = torch.zeros(1024, 768, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(768, 768, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(768, device="cuda")
    y = torch.zeros(1024, 768, device="cuda", dtype=torch.bfloat16)
    ld = _lib.FavLinearDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 1024, 768, 768, 0)   # 48 tiles
    assert lib.fav_op_linear_streamk(C.byref(ld), None) != 0
