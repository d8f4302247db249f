"""Per-kernel parity: every HIP kernel, called through the C ABI (fav_op_*), against
the CPU oracle on the same seeded inputs.  One kernel = one rounding point, so the
bar is tight: integer/bf16-exact where the arithmetic is order-free (im2col, pools,
dropout masks, labels), and at most one bf16 ulp on well under 1% of elements for
the MFMA convolution (fp32 summation order is the only difference)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import _lib  # noqa: E402
from oracle import fav_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return _lib.load()


def dev_bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda().to(torch.bfloat16).contiguous()


def host_f32(t):
    return t.to(torch.float32).cpu().numpy()


def drop_desc(site=-1, thr=0, scale=1.0, seed=0, v0=0, n_img=1, first=0):
    return _lib.FavDropoutDesc(site, thr, scale, seed, v0, n_img, first)


def run_conv(lib, x, w, b, res=None, stride=1, pad=0, relu=1, out_f32=0, math_mode=0, drop=None):
    n, H, W, cin = x.shape
    cout, kh, kw, _ = w.shape
    ho, wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    xd, wd = dev_bf16(x), dev_bf16(w)
    bd = torch.from_numpy(b.astype(np.float32)).cuda()
    rd = dev_bf16(res) if res is not None else None
    y = torch.empty((n, ho, wo, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    d = _lib.FavConvDesc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr() if rd is not None else None,
                         y.data_ptr(), n, H, W, cin, cout, kh, kw, stride, pad, relu, out_f32, math_mode,
                         drop or drop_desc())
    _lib.check(lib.fav_op_conv2d(C.byref(d), None))
    torch.cuda.synchronize()
    return host_f32(y)


def assert_one_ulp(got, ref, frac=0.01):
    d = np.abs(got - ref)
    assert (d > 0).mean() < frac, f"{(d > 0).mean():.4f} of elements differ"
    assert np.all(d <= np.maximum(np.abs(got), np.abs(ref)) * 2.0 ** -7 + 1e-5), d.max()


CONV_CASES = [
    # cin, cout, k, stride, pad, H, W, n, residual
    (64, 64, 1, 1, 0, 14, 14, 3, False),
    (64, 64, 3, 1, 1, 13, 9, 2, False),
    (64, 128, 3, 2, 1, 15, 15, 2, False),
    (128, 256, 1, 1, 0, 7, 7, 5, True),
    (256, 512, 1, 2, 0, 14, 14, 2, False),
    (128, 128, 3, 1, 1, 28, 28, 2, True),
    (512, 128, 1, 1, 0, 5, 5, 1, False),      # M=25: a single partial tile
    (192, 64, 1, 1, 0, 16, 16, 2, False),     # the stem GEMM shape (K=192)
    (64, 64, 3, 1, 1, 56, 56, 1, True),       # 25 tiles, tail tile of 64 rows
    (64, 64, 3, 1, 1, 56, 56, 6, True),       # M=18816 -> 256-row tiles (8 waves), BN=64, ragged last tile
    (128, 128, 3, 2, 1, 57, 57, 21, True),    # M=17661, stride 2, odd size
    (128, 256, 3, 1, 1, 30, 30, 19, False),   # M=17100, Cout 256, no residual -> the 256x256 tile (8 waves)
    (512, 256, 1, 1, 0, 30, 30, 19, False),   # 1x1 with K=512 on the 256x256 tile
]


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,n,use_res", CONV_CASES)
def test_conv2d_bf16_vs_oracle(lib, cin, cout, k, stride, pad, H, W, n, use_res):
    rng = np.random.default_rng(cin * 1000 + cout + k)
    x = O.bf16_round(np.maximum(rng.standard_normal((n, H, W, cin)), -0.5).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    L = O.ConvLayer(cout, cin, k, k, stride, pad, w, b)
    acc = O.conv_acc(x, L)
    res = O.bf16_round(rng.standard_normal(acc.shape).astype(np.float32)) if use_res else None
    ref = O.epilogue(acc, b, res=res, relu=True)
    got = run_conv(lib, x, w, b, res, stride, pad, relu=1)
    assert got.shape == ref.shape
    assert_one_ulp(got, ref)
    # no-ReLU variant (downsample path) and fp32 output (classifier)
    got2 = run_conv(lib, x, w, b, None, stride, pad, relu=0)
    assert_one_ulp(got2, O.epilogue(acc, b, relu=False))
    got3 = run_conv(lib, x, w, b, None, stride, pad, relu=0, out_f32=1)
    np.testing.assert_allclose(got3, acc + b, rtol=2e-5, atol=2e-5)


def test_conv2d_exact_input_catches_transposes(lib):
    """Integer-valued operands (every product and sum exact in fp32) with asymmetric
    weights: any swapped row/column/k mapping in the MFMA fragment code shows up as an
    exact mismatch, independent of rounding."""
    rng = np.random.default_rng(7)
    n, H, W, cin, cout = 1, 12, 12, 128, 256
    x = rng.integers(-3, 4, (n, H, W, cin)).astype(np.float32)
    w = rng.integers(-2, 3, (cout, 3, 3, cin)).astype(np.float32)
    b = rng.integers(-4, 5, cout).astype(np.float32)
    L = O.ConvLayer(cout, cin, 3, 3, 1, 1, w, b)
    ref = O.conv_acc(x, L) + b
    for mode in (0, 1):
        got = run_conv(lib, x, w, b, None, 1, 1, relu=0, out_f32=1, math_mode=mode)
        assert np.array_equal(got, ref), f"math_mode {mode}: max diff {np.abs(got - ref).max()}"


def test_conv2d_fused_dropout_mask_is_philox(lib):
    rng = np.random.default_rng(9)
    n, H, W, cin, cout = 6, 7, 7, 64, 128
    x = O.bf16_round(np.abs(rng.standard_normal((n, H, W, cin))).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((cout, 1, 1, cin)) * 0.2).astype(np.float32))
    b = np.full(cout, 0.5, np.float32)
    L = O.ConvLayer(cout, cin, 1, 1, 1, 0, w, b)
    thr = O.dropout_threshold(0.25)
    scale = O.dropout_scale(thr)
    # rows are virtual frames v0..v0+5 of a 4-frame batch: v=5,6,7 -> t=1 (img 1..3), v=8,9,10 -> t=2
    v0, n_img, first, seed, site = 5, 4, 100, 0x1234567890ABCDEF, 3
    keep = np.empty((n, H * W * cout), bool)
    for i in range(n):
        v = v0 + i
        keep[i] = O.dropout_keep(seed, v // n_img, site, np.array([first + v % n_img]), H * W * cout, thr)[0]
    ref = O.epilogue(O.conv_acc(x, L), b, relu=True, keep=keep, scale=scale)
    got = run_conv(lib, x, w, b, None, 1, 0, relu=1, drop=drop_desc(site, thr, float(scale), seed, v0, n_img, first))
    assert np.array_equal(got == 0, ref == 0), "dropout mask differs from the oracle's Philox mask"
    live = O.epilogue(O.conv_acc(x, L), b, relu=True) > 0          # survivors of the ReLU
    assert abs((got[live] == 0).mean() - 0.25) < 0.02
    assert_one_ulp(got, ref)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("k,stride,pad,kpad,H,W", [(7, 2, 3, 192, 32, 40), (3, 1, 1, 64, 16, 16),
                                                   (16, 16, 0, 768, 64, 48),     # ViT patch embedding: the 16-byte-piece path
                                                   (16, 16, 0, 768, 32, 40),     # u8 rows of 120 bytes: the scalar path for u8, pieces for fp32
                                                   (4, 4, 0, 64, 16, 36)])
def test_stem_im2col_exact(lib, layout, k, stride, pad, kpad, H, W):
    rng = np.random.default_rng(3)
    n = 3
    if layout == 0:
        img = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    else:
        img = rng.random((n, H, W, 3), dtype=np.float32)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    istd = O.inv_std32(std)
    xn = O.normalize_input(img, mean, istd)
    cols, ho, wo = O._im2col(xn, k, k, stride, pad)
    ref = np.zeros((cols.shape[0], kpad), np.float32)
    ref[:, :cols.shape[1]] = cols
    out = torch.empty((n * ho * wo, kpad), dtype=torch.bfloat16, device="cuda")
    m3 = (C.c_float * 3)(*mean)
    i3 = (C.c_float * 3)(*[float(v) for v in istd])
    _lib.check(lib.fav_op_stem_im2col(torch.from_numpy(img).cuda().data_ptr(), layout, n, H, W, k, k, stride, pad, kpad,
                                      m3, i3, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(host_f32(out), ref)


def test_maxpool_exact(lib):
    rng = np.random.default_rng(4)
    for (n, H, W, Cc) in [(2, 16, 16, 64), (1, 15, 9, 128), (3, 112, 112, 64)]:
        x = O.bf16_round(rng.standard_normal((n, H, W, Cc)).astype(np.float32))
        ref = O.maxpool3x3s2(x)
        y = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.fav_op_maxpool3x3s2(dev_bf16(x).data_ptr(), y.data_ptr(), n, H, W, Cc, None))
        torch.cuda.synchronize()
        assert np.array_equal(host_f32(y), ref)


def test_avgpool_exact_and_dropout(lib):
    rng = np.random.default_rng(5)
    n, HW, Cc = 5, 49, 512
    x = O.bf16_round(np.abs(rng.standard_normal((n, 7, 7, Cc))).astype(np.float32))
    ref = O.bf16_round(O.global_avgpool(x))
    y = torch.empty((n, Cc), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_avgpool(dev_bf16(x).data_ptr(), y.data_ptr(), n, HW, Cc, None, None))
    torch.cuda.synchronize()
    assert np.array_equal(host_f32(y), ref)  # same sequential fp32 order -> bit exact
    thr = O.dropout_threshold(0.5)
    scale = O.dropout_scale(thr)
    d = drop_desc(16, thr, float(scale), 77, 2, 5, 40)   # v = 2..6 of a 5-frame batch
    keep = np.stack([O.dropout_keep(77, (2 + i) // 5, 16, np.array([40 + (2 + i) % 5]), Cc, thr)[0] for i in range(n)])
    ref2 = O.bf16_round(np.where(keep, O.global_avgpool(x) * scale, 0).astype(np.float32))
    _lib.check(lib.fav_op_avgpool(dev_bf16(x).data_ptr(), y.data_ptr(), n, HW, Cc, C.byref(d), None))
    torch.cuda.synchronize()
    assert np.array_equal(host_f32(y), ref2)


def test_entry_dropout_exact(lib):
    rng = np.random.default_rng(6)
    n_img, E, n_out, v0 = 3, 7 * 7 * 64, 7, 2
    x = O.bf16_round(np.abs(rng.standard_normal((n_img, E))).astype(np.float32))
    thr = O.dropout_threshold(0.1)
    scale = O.dropout_scale(thr)
    d = drop_desc(2, thr, float(scale), 5, v0, n_img, 9)
    out = torch.empty((n_out, E), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_entry_dropout(dev_bf16(x).data_ptr(), out.data_ptr(), E, n_out, C.byref(d), None))
    torch.cuda.synchronize()
    ref = np.empty((n_out, E), np.float32)
    for i in range(n_out):
        v = v0 + i
        keep = O.dropout_keep(5, v // n_img, 2, np.array([9 + v % n_img]), E, thr)[0]
        ref[i] = O.bf16_round(np.where(keep, x[v % n_img] * scale, 0).astype(np.float32))
    assert np.array_equal(host_f32(out), ref)


@pytest.mark.parametrize("n_img,H,W,v0,n_out", [(3, 7, 9, 2, 7),        # partial first and last sample, tiles span frames
                                                 (5, 14, 14, 0, 15),     # three whole samples, ragged last tile (M = 980)
                                                 (2, 56, 56, 1, 2),      # layer-1 frames, window inside the samples
                                                 (9, 5, 3, 4, 1)])       # a single virtual frame
def test_entry_reduce_bitwise_vs_separate_launches_and_oracle(lib, n_img, H, W, v0, n_out):
    """Entry dropout + the 1x1 reduce behind it in one launch (entry_reduce_kernel): y and t1 bit-identical to
    fav_op_entry_dropout followed by fav_op_conv2d, and to the oracle (Philox masks, MFMA-model accumulation)."""
    rng = np.random.default_rng(n_img * 100 + H + v0)
    Cc, nred, HW = 256, 64, H * W
    x = O.bf16_round(np.maximum(rng.standard_normal((n_img, H, W, Cc)) * np.exp2(rng.integers(-2, 3, (n_img, H, W, Cc))), -0.1).astype(np.float32))
    wa = O.bf16_round((rng.standard_normal((nred, 1, 1, Cc)) * np.sqrt(2.0 / Cc)).astype(np.float32))
    ba = (rng.standard_normal(nred) * 0.2).astype(np.float32)
    thr = O.dropout_threshold(0.1)
    scale = O.dropout_scale(thr)
    d = drop_desc(1, thr, float(scale), 77, v0, n_img, 40)
    xd, wd, bd = dev_bf16(x), dev_bf16(wa), torch.from_numpy(ba).cuda()
    y = torch.zeros((n_out, H, W, Cc), dtype=torch.bfloat16, device="cuda")
    t1 = torch.zeros((n_out, H, W, nred), dtype=torch.bfloat16, device="cuda")
    guard = 4096                                                   # rows outside the window must stay untouched
    ybig = torch.full((n_out * HW * Cc + 2 * guard,), 3.0, dtype=torch.bfloat16, device="cuda")
    tbig = torch.full((n_out * HW * nred + 2 * guard,), 3.0, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_entry_reduce(xd.data_ptr(), ybig.data_ptr() + 2 * guard, wd.data_ptr(), bd.data_ptr(), tbig.data_ptr() + 2 * guard,
                                       Cc, nred, HW, n_out, C.byref(d), None))
    torch.cuda.synchronize()
    for big in (ybig, tbig):
        assert bool((big[:guard] == 3.0).all()) and bool((big[-guard:] == 3.0).all())
    y = host_f32(ybig[guard:-guard].reshape(n_out, H, W, Cc))
    t1 = host_f32(tbig[guard:-guard].reshape(n_out, H, W, nred))
    # (a) the separate launches
    y2 = torch.empty((n_out, HW * Cc), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_entry_dropout(xd.data_ptr(), y2.data_ptr(), HW * Cc, n_out, C.byref(d), None))
    torch.cuda.synchronize()
    y_ref = host_f32(y2).reshape(n_out, H, W, Cc)
    t1_ref = run_conv(lib, y_ref, wa, ba, None, 1, 0, relu=1)
    assert np.array_equal(y, y_ref), f"y: {np.mean(y != y_ref):.5f} of elements differ"
    assert np.array_equal(t1, t1_ref), f"t1: {np.mean(t1 != t1_ref):.5f} of elements differ"
    # (b) the oracle
    oy = np.empty_like(y)
    for i in range(n_out):
        v = v0 + i
        keep = O.dropout_keep(77, v // n_img, 1, np.array([40 + v % n_img]), HW * Cc, thr)[0].reshape(H, W, Cc)
        oy[i] = O.bf16_round(np.where(keep, x[v % n_img] * scale, 0).astype(np.float32))
    assert np.array_equal(y, oy)
    assert np.array_equal(t1, O.epilogue(O.conv_acc_exact(oy, wa, 1, 1, 1, 0, mode="mfma"), ba))


def test_entry_reduce_rejects_unsupported_shapes(lib):
    x = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    d = drop_desc(1, 26, 1.1, 7, 0, 1, 0)
    assert lib.fav_op_entry_reduce(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 128, 64, 4, 1, C.byref(d), None) == 1


@pytest.mark.parametrize("T,n,Cc,ld", [(1, 5, 1000, 1024), (30, 9, 1000, 1024), (3, 4, 10, 64), (7, 3, 257, 320)])
def test_head_vs_oracle(lib, T, n, Cc, ld):
    rng = np.random.default_rng(T * 100 + Cc)
    lg = np.zeros((T, n, ld), np.float32)
    lg[:, :, :Cc] = (rng.standard_normal((T, n, Cc)) * 4).astype(np.float32)
    lg[:, :, Cc:] = 1e9  # padding columns must be ignored
    for kind in (0, 1):
        rl, rc, pbar = O.confidence_head(lg[:, :, :Cc], temperature=1.3, kind=kind)
        rf, rs = O.failure_detect(rc, 0.4)
        ld_t = torch.from_numpy(lg).cuda()
        labels = torch.empty(n, dtype=torch.int32, device="cuda")
        conf = torch.empty(n, dtype=torch.float32, device="cuda")
        fail = torch.empty(n, dtype=torch.uint8, device="cuda")
        score = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(lib.fav_op_head(ld_t.data_ptr(), T, n, Cc, ld, 1.3, kind, 0.4, labels.data_ptr(), conf.data_ptr(),
                                   fail.data_ptr(), score.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(labels.cpu().numpy(), rl)
        np.testing.assert_allclose(conf.cpu().numpy(), rc, rtol=0, atol=2e-6)
        np.testing.assert_allclose(score.cpu().numpy(), rs, rtol=0, atol=2e-6)
        near = np.abs(rc - 0.4) < 1e-5
        assert np.array_equal(fail.cpu().numpy()[~near], rf[~near])


def test_head_tie_breaks_to_lowest_index(lib):
    lg = np.zeros((1, 2, 64), np.float32)
    lg[0, 0, [40, 7, 23]] = 2.0
    lg[0, 1, :] = 0.0
    labels = torch.empty(2, dtype=torch.int32, device="cuda")
    conf = torch.empty(2, dtype=torch.float32, device="cuda")
    _lib.check(lib.fav_op_head(torch.from_numpy(lg).cuda().data_ptr(), 1, 2, 50, 64, 1.0, 0, 0.5, labels.data_ptr(),
                               conf.data_ptr(), None, None, None))
    torch.cuda.synchronize()
    assert labels.cpu().tolist() == [7, 0]


# Launches large enough for the 256 x 256 ping-pong tile against the SAME operation on 32-frame slices, which take the 128-row tiles:
# every tile kernel implements one k order, so the bits must agree whatever the tile and the grid - a size-independent property at a
# size no CPU oracle reaches (and the check any other schedule of that tile has to pass: tools/experiments/r4_persistent_big_tile.diff).
BIG_TILE_CASES = [
    # cin, cout, k, H, W, frames, residual, relu, dropout
    (512, 512, 1, 14, 14, 1024, True, 1, True),
    (512, 512, 1, 14, 14, 1023, True, 1, False),      # a partial last tile, tile count not a multiple of the grid
    (256, 256, 3, 14, 14, 1024, False, 1, True),
    (768, 3072, 1, 197, 1, 512, False, 2, False),     # ViT fc1 + GELU
    (768, 768, 1, 197, 1, 700, True, 0, False),       # ViT proj + residual
]


@pytest.mark.parametrize("cin,cout,k,H,W,n,use_res,relu,use_drop", BIG_TILE_CASES)
def test_big_tile_matches_sliced_launches(lib, cin, cout, k, H, W, n, use_res, relu, use_drop):
    g = torch.Generator(device="cuda").manual_seed(cin + cout + k + n)
    pad = k // 2
    x = (torch.randn((n, H, W, cin), device="cuda", generator=g) * 0.7).to(torch.bfloat16)
    w = (torch.randn((cout, k, k, cin), device="cuda", generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.bfloat16)
    b = torch.randn((cout,), device="cuda", generator=g) * 0.2
    res = torch.randn((n, H, W, cout), device="cuda", generator=g).to(torch.bfloat16) if use_res else None
    y = torch.zeros((n, H, W, cout), dtype=torch.bfloat16, device="cuda")
    y2 = torch.zeros_like(y)

    def launch(lo, hi, out):
        dd = drop_desc(3, 26, 1.0 / (1 - 26 / 256), 77, lo, n, 5) if use_drop else drop_desc()
        d = _lib.FavConvDesc(x[lo:hi].data_ptr(), w.data_ptr(), b.data_ptr(), res[lo:hi].data_ptr() if use_res else None,
                             out[lo:hi].data_ptr(), hi - lo, H, W, cin, cout, k, k, 1, pad, relu, 0, 0, dd)
        _lib.check(lib.fav_op_conv2d(C.byref(d), None))

    launch(0, n, y)
    for lo in range(0, n, 32):
        launch(lo, min(n, lo + 32), y2)
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), \
        f"{(y.view(torch.int16) != y2.view(torch.int16)).float().mean().item():.6f} of elements differ"
    assert y.float().abs().sum().item() > 0
