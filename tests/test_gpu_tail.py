"""bottleneck_tail_kernel (conv_b 3x3 -> conv_c 1x1 + residual + dropout -> the next block's conv_a 1x1 in one
launch) against (a) the three separate fav_op_conv2d launches it replaces and (b) the CPU oracle with the bit-exact
model of v_mfma_f32_16x16x32_bf16: every output element bit-identical - same k order, same rounding points, same
Philox draws.  Shapes cover both channel widths, every fused combination the executor builds, ragged last tiles,
frames smaller than a tile and non-square frames."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import _lib  # noqa: E402
from oracle import fav_oracle as O  # noqa: E402
from test_gpu_ops import dev_bf16, drop_desc, host_f32, run_conv  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def run_tail(lib, x, wb, bb, wc, bc, res, wa, ba, drop=None):
    n, H, W, cmid = x.shape
    cout = wc.shape[0]
    nred = wa.shape[0] if wa is not None else 0
    keep = [dev_bf16(x), dev_bf16(wc), torch.from_numpy(bc).cuda(), dev_bf16(res)]
    y = torch.zeros((n, H, W, cout), dtype=torch.bfloat16, device="cuda")
    t1n = torch.zeros((n, H, W, max(nred, 1)), dtype=torch.bfloat16, device="cuda")
    wbd = dev_bf16(wb) if wb is not None else None
    bbd = torch.from_numpy(bb).cuda() if wb is not None else None
    wad = dev_bf16(wa) if wa is not None else None
    bad = torch.from_numpy(ba).cuda() if wa is not None else None
    d = _lib.FavTailDesc(keep[0].data_ptr(), wbd.data_ptr() if wbd is not None else None,
                         bbd.data_ptr() if bbd is not None else None, keep[1].data_ptr(), keep[2].data_ptr(),
                         keep[3].data_ptr(), y.data_ptr(), wad.data_ptr() if wad is not None else None,
                         bad.data_ptr() if bad is not None else None, t1n.data_ptr() if wad is not None else None,
                         n, H, W, cmid, nred, drop or drop_desc())
    _lib.check(lib.fav_op_bottleneck_tail(C.byref(d), None))
    torch.cuda.synchronize()
    return host_f32(y), (host_f32(t1n) if wa is not None else None)


CASES = [
    # cmid, nred, has3x3, H, W, n
    (64, 64, True, 56, 56, 2),      # layer 1 inner boundary; M = 6272 = 49 tiles
    (64, 128, True, 56, 56, 1),     # layer 1 -> layer 2 (the next block reduces to 128)
    (64, 0, True, 20, 12, 3),       # no next block; ragged last tile (M = 720)
    (64, 64, True, 7, 9, 37),       # frames much smaller than a tile; M = 2331 (ragged)
    (64, 64, True, 60, 80, 1),      # the 240x320 seam at layer 1 (H != W)
    (128, 128, True, 28, 28, 3),    # layer 2 inner boundary
    (128, 0, True, 28, 28, 2),      # last block of the high-resolution group
    (128, 128, False, 28, 28, 3),   # the stride-2 block: its 3x3 stays a separate launch
    (64, 64, False, 14, 10, 5),
    (128, 128, True, 9, 11, 7),
    (256, 0, False, 14, 14, 11),    # layer 3's expanding 1x1 alone (wide tail: weight buffers reuse the T2 region); M = 2156
    (256, 0, False, 7, 9, 3),
    (256, 256, False, 14, 14, 11),  # layer 3 inner boundary: expand + the next block's reduce, 8 waves x 16 rows
    (256, 256, False, 5, 7, 9),
    (512, 0, False, 7, 7, 45),      # layer 4's expanding 1x1 alone (single Wc buffer of 64 KB); M = 2205 (ragged)
    (512, 0, False, 8, 10, 3),
    (256, 0, True, 14, 14, 11),     # layer 3: conv_b as the generic 256x256x64 loop + conv_c in one launch; M = 2156 (ragged)
    (256, 0, True, 7, 9, 5),        # frames much smaller than a tile
    (256, 0, True, 15, 20, 2),      # the 240x320 seam at layer 3 (H != W)
]


@pytest.mark.parametrize("cmid,nred,has3x3,H,W,n", CASES)
@pytest.mark.parametrize("with_drop", [False, True])
def test_tail_bitwise_vs_separate_launches_and_oracle(lib, cmid, nred, has3x3, H, W, n, with_drop):
    rng = np.random.default_rng(cmid * 31 + nred * 7 + H * W + n + int(has3x3))
    cout = 4 * cmid
    x = O.bf16_round(np.maximum(rng.standard_normal((n, H, W, cmid)) * np.exp2(rng.integers(-2, 3, (n, H, W, cmid))), -0.2).astype(np.float32))
    wb = O.bf16_round((rng.standard_normal((cmid, 3, 3, cmid)) * np.sqrt(2.0 / (9 * cmid))).astype(np.float32)) if has3x3 else None
    bb = (rng.standard_normal(cmid) * 0.2).astype(np.float32) if has3x3 else None
    wc = O.bf16_round((rng.standard_normal((cout, 1, 1, cmid)) * np.sqrt(1.0 / cmid)).astype(np.float32))
    bc = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    res = O.bf16_round(np.maximum(rng.standard_normal((n, H, W, cout)), 0).astype(np.float32))
    wa = O.bf16_round((rng.standard_normal((nred, 1, 1, cout)) * np.sqrt(2.0 / cout)).astype(np.float32)) if nred else None
    ba = (rng.standard_normal(nred) * 0.2).astype(np.float32) if nred else None
    thr = 26
    dd = drop_desc(site=5, thr=thr, scale=float(O.dropout_scale(thr)), seed=0x1234567ABC, v0=3, n_img=n + 1, first=40) if with_drop else None
    # (a) the separate launches
    t2 = run_conv(lib, x, wb, bb, None, 1, 1, relu=1) if has3x3 else x
    y_ref = run_conv(lib, t2, wc, bc, res, 1, 0, relu=1, drop=dd)
    t1n_ref = run_conv(lib, y_ref, wa, ba, None, 1, 0, relu=1) if nred else None
    y, t1n = run_tail(lib, x, wb, bb, wc, bc, res, wa, ba, drop=dd)
    assert np.array_equal(y, y_ref), f"Y: {np.mean(y != y_ref):.5f} of elements differ from the separate launches"
    if nred:
        assert np.array_equal(t1n, t1n_ref), f"t1': {np.mean(t1n != t1n_ref):.5f} of elements differ"
    # (b) the oracle (MFMA model), small cases only
    if n * H * W <= 8000:
        keep = None
        if with_drop:
            # virtual frame v = v0 + i: sample t = v // n_img, frame (v % n_img) + first
            v = 3 + np.arange(n)
            t, img = v // (n + 1), v % (n + 1) + 40
            keep = np.stack([O.dropout_keep(0x1234567ABC, int(tt), 5, np.array([ii]), H * W * cout, thr)[0] for tt, ii in zip(t, img)])
        o2 = O.epilogue(O.conv_acc_exact(x, wb, 3, 3, 1, 1, mode="mfma"), bb) if has3x3 else x
        oy = O.epilogue(O.conv_acc_exact(o2, wc, 1, 1, 1, 0, mode="mfma"), bc, res=res, relu=True, keep=keep, scale=O.dropout_scale(thr))
        assert np.array_equal(y, oy), f"Y vs oracle: {np.mean(y != oy):.5f} differ"
        if nred:
            assert np.array_equal(t1n, O.epilogue(O.conv_acc_exact(oy, wa, 1, 1, 1, 0, mode="mfma"), ba))


def test_tail_rejects_unsupported_shapes(lib):
    x = torch.zeros((1, 8, 8, 96), dtype=torch.bfloat16, device="cuda")
    d = _lib.FavTailDesc(x.data_ptr(), None, None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, None,
                         1, 8, 8, 96, 0, drop_desc())
    assert lib.fav_op_bottleneck_tail(C.byref(d), None) == 1      # FAV_ERR_INVALID_ARG, nothing launched


@pytest.mark.parametrize("stride,H,W,n", [(2, 56, 56, 6), (2, 30, 26, 41), (1, 28, 28, 7)])
def test_projection_shortcut_on_the_row_owning_kernel(lib, stride, H, W, n):
    """1x1 / stride s, 256 -> 512, no residual, no ReLU (a stage's projection shortcut) takes the tail kernel's
    row-owning path when the launch is large enough: bit-identical to the MFMA-model oracle, ragged last tile included."""
    rng = np.random.default_rng(stride * 100 + H + n)
    x = O.bf16_round((rng.standard_normal((n, H, W, 256)) * np.exp2(rng.integers(-2, 3, (n, H, W, 256)))).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((512, 1, 1, 256)) * np.sqrt(1.0 / 256)).astype(np.float32))
    b = (rng.standard_normal(512) * 0.2).astype(np.float32)
    got = run_conv(lib, x, w, b, None, stride, 0, relu=0)
    ref = O.epilogue(O.conv_acc_exact(x, w, 1, 1, stride, 0, mode="mfma"), b, relu=False)
    assert got.shape == ref.shape and got.shape[0] * got.shape[1] * got.shape[2] >= 4096
    assert np.array_equal(got, ref), f"{np.mean(got != ref):.5f} of elements differ"


@pytest.mark.parametrize("stride,H,W,n", [(2, 28, 28, 672), (1, 14, 14, 700)])
def test_wide_projection_shortcut_on_the_row_owning_kernel(lib, stride, H, W, n):
    """1x1 / stride s, 512 -> 1024 (layer 3's projection shortcut) takes the 8-wave row-owning kernel once the launch
    brings two blocks per CU (M >= 131 072 output pixels).  Tensors are made on the device (the input is 0.5 GB);
    the whole output is compared with the generic kernel (its fp32 output, rounded here: same accumulation, same
    rounding point) and the first / last frames with the MFMA-model oracle."""
    g = torch.Generator(device="cuda").manual_seed(stride * 1000 + n)
    x = (torch.randn((n, H, W, 512), device="cuda", generator=g) * 0.7).to(torch.bfloat16)
    w = (torch.randn((1024, 1, 1, 512), device="cuda", generator=g) * (1.0 / 512) ** 0.5).to(torch.bfloat16)
    b = torch.randn(1024, device="cuda", generator=g) * 0.2
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert n * Ho * Wo >= 131072
    y = torch.zeros((n, Ho, Wo, 1024), dtype=torch.bfloat16, device="cuda")
    yf = torch.zeros((n, Ho, Wo, 1024), dtype=torch.float32, device="cuda")
    for out, f32 in ((y, 0), (yf, 1)):
        d = _lib.FavConvDesc(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), n, H, W, 512, 1024, 1, 1, stride, 0, 0, f32, 0, drop_desc())
        _lib.check(lib.fav_op_conv2d(C.byref(d), None))
    torch.cuda.synchronize()
    assert torch.equal(y, yf.to(torch.bfloat16)), "row-owning kernel differs from the generic kernel"
    for sl in (slice(0, 3), slice(n - 3, n)):
        xs, ws = x[sl].float().cpu().numpy(), w.float().cpu().numpy()
        ref = O.epilogue(O.conv_acc_exact(xs, ws, 1, 1, stride, 0, mode="mfma"), b.cpu().numpy(), relu=False)
        assert np.array_equal(y[sl].float().cpu().numpy(), ref)


@pytest.mark.parametrize("H,W,n_img,v0,n_out", [(56, 56, 3, 0, 9), (20, 12, 4, 2, 9), (7, 9, 5, 13, 37), (60, 80, 2, 1, 3),
                                                (16, 16, 2, 4, 6), (8, 16, 3, 0, 6), (56, 56, 4, 0, 12)])   # whole samples of whole tiles: sample-minor tile order
def test_tail_entry_residual_recomputes_the_dropped_copies(lib, H, W, n_img, v0, n_out):
    """res_entry: the tail behind the entry dropout of an MC-Dropout suffix takes its residual from the CACHED prefix
    output x0 [n_img] - virtual frame v reads frame v % n_img and applies the entry site's mask in the epilogue - and gives
    the bits of the same tail fed with the stored dropped copies (fav_op_entry_reduce with y), for chunks of virtual
    frames that start inside a sample, wrap around the cached frames and end in a ragged tile; and entry_reduce with
    y = NULL still writes the same t1."""
    rng = np.random.default_rng(H * W + n_img * 7 + v0)
    cmid, cout, nred = 64, 256, 64
    thr, seed, first, site_e, site_o = 26, 0x1234567ABC, 40, 2, 3
    x0 = O.bf16_round(np.maximum(rng.standard_normal((n_img, H, W, cout)), 0).astype(np.float32))
    wa0 = O.bf16_round((rng.standard_normal((nred, 1, 1, cout)) * np.sqrt(2.0 / cout)).astype(np.float32))
    ba0 = (rng.standard_normal(nred) * 0.2).astype(np.float32)
    wb = O.bf16_round((rng.standard_normal((cmid, 3, 3, cmid)) * np.sqrt(2.0 / (9 * cmid))).astype(np.float32))
    bb = (rng.standard_normal(cmid) * 0.2).astype(np.float32)
    wc = O.bf16_round((rng.standard_normal((cout, 1, 1, cmid)) * np.sqrt(1.0 / cmid)).astype(np.float32))
    bc = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    wa = O.bf16_round((rng.standard_normal((nred, 1, 1, cout)) * np.sqrt(2.0 / cout)).astype(np.float32))
    ba = (rng.standard_normal(nred) * 0.2).astype(np.float32)
    scale = float(O.dropout_scale(thr))
    de = drop_desc(site=site_e, thr=thr, scale=scale, seed=seed, v0=v0, n_img=n_img, first=first)
    do = drop_desc(site=site_o, thr=thr, scale=scale, seed=seed, v0=v0, n_img=n_img, first=first)
    x0d, wa0d, ba0d = dev_bf16(x0), dev_bf16(wa0.reshape(nred, cout)), torch.from_numpy(ba0).cuda()
    # the stored form: entry dropout + reduce -> y0 (dropped copies), t1
    y0 = torch.zeros((n_out, H, W, cout), dtype=torch.bfloat16, device="cuda")
    t1 = torch.zeros((n_out, H, W, nred), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_entry_reduce(x0d.data_ptr(), y0.data_ptr(), wa0d.data_ptr(), ba0d.data_ptr(), t1.data_ptr(), cout, nred,
                                       H * W, n_out, C.byref(de), None))
    t1b = torch.zeros_like(t1)
    _lib.check(lib.fav_op_entry_reduce(x0d.data_ptr(), None, wa0d.data_ptr(), ba0d.data_ptr(), t1b.data_ptr(), cout, nred,
                                       H * W, n_out, C.byref(de), None))
    torch.cuda.synchronize()
    assert torch.equal(t1, t1b)
    y_ref, t1n_ref = run_tail(lib, host_f32(t1), wb, bb, wc, bc, host_f32(y0), wa, ba, drop=do)
    # the recomputed form
    keep = [t1, dev_bf16(wb), torch.from_numpy(bb).cuda(), dev_bf16(wc), torch.from_numpy(bc).cuda(), dev_bf16(wa), torch.from_numpy(ba).cuda()]
    y = torch.zeros((n_out, H, W, cout), dtype=torch.bfloat16, device="cuda")
    t1n = torch.zeros((n_out, H, W, nred), dtype=torch.bfloat16, device="cuda")
    d = _lib.FavTailDesc(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                         x0d.data_ptr(), y.data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), t1n.data_ptr(),
                         n_out, H, W, cmid, nred, do, 1, site_e)
    _lib.check(lib.fav_op_bottleneck_tail(C.byref(d), None))
    torch.cuda.synchronize()
    assert np.array_equal(host_f32(y), y_ref), f"Y: {np.mean(host_f32(y) != y_ref):.5f} of elements differ"
    assert np.array_equal(host_f32(t1n), t1n_ref)
    # and the dropped copies really are dropout_{site_e}(x0[v % n_img]) as the oracle draws it
    v = v0 + np.arange(n_out)
    keepm = np.stack([O.dropout_keep(seed, int(tt), site_e, np.array([ii]), H * W * cout, thr)[0]
                      for tt, ii in zip(v // n_img, v % n_img + first)]).reshape(n_out, H, W, cout)
    exp = O.bf16_round(np.where(keepm, x0[v % n_img] * np.float32(scale), 0).astype(np.float32))
    assert np.array_equal(host_f32(y0), exp)
    # unsupported uses are refused
    bad = _lib.FavTailDesc(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(),
                           x0d.data_ptr(), y.data_ptr(), keep[5].data_ptr(), keep[6].data_ptr(), t1n.data_ptr(),
                           n_out, H, W, cmid, nred, drop_desc(), 1, site_e)
    assert lib.fav_op_bottleneck_tail(C.byref(bad), None) != 0
