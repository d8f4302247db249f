"""Bit-for-bit parity in FAV_MATH_F32_EXACT mode.

In this mode the HIP kernel feeds the same bf16 operands, tiles and epilogue to the
fp32-input MFMA, whose accumulation is a k-ordered fmaf chain; oracle/fav_exact.c
restates that chain on the CPU.  Everything else on the path (normalisation, im2col,
bias/residual/ReLU/dropout epilogue, pools, Philox masks) is order-free fp32 or
integer arithmetic, so the LOGITS must be bit-identical and labels exactly equal, for
every architecture, dropout policy and chunking.  The production bf16 mode differs
from this mode only in the MFMA instruction issued (same operands, same k-tiles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from failure_aware_vision_amd import Backend, _lib, synth, weights  # noqa: E402
from oracle import fav_oracle as O  # noqa: E402
from test_gpu_ops import run_conv  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,n", [
    (64, 64, 1, 1, 0, 14, 14, 3), (64, 128, 3, 2, 1, 15, 15, 2), (128, 256, 3, 1, 1, 9, 9, 2),
    (512, 128, 1, 1, 0, 7, 7, 3), (192, 64, 1, 1, 0, 16, 16, 2), (256, 512, 1, 2, 0, 14, 14, 1),
    (64, 128, 3, 1, 1, 40, 44, 10),   # M=17600
    (64, 256, 3, 1, 1, 40, 44, 10),   # M=17600, Cout 256 -> the 256x256 tile (8 waves)
])
def test_conv_exact_bitwise(lib, cin, cout, k, stride, pad, H, W, n):
    rng = np.random.default_rng(cin + cout + k)
    x = O.bf16_round((rng.standard_normal((n, H, W, cin)) * np.exp2(rng.integers(-3, 4, (n, H, W, cin)))).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((cout, k, k, cin)) * 0.1).astype(np.float32))
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    acc = O.conv_acc_exact(x, w, k, k, stride, pad)
    got = run_conv(lib, x, w, b, None, stride, pad, relu=0, out_f32=1, math_mode=1)
    assert np.array_equal(got, acc + b)
    res = O.bf16_round(rng.standard_normal(acc.shape).astype(np.float32))
    got2 = run_conv(lib, x, w, b, res, stride, pad, relu=1, math_mode=1)
    assert np.array_equal(got2, O.epilogue(acc, b, res=res, relu=True))


@pytest.mark.parametrize("c,H,W,n", [
    (64, 56, 56, 2),     # M = 6272: 24.5 tiles of 256 pixels (ragged last tile)
    (64, 60, 80, 1),     # H != W (the 240x320 seam at layer 1)
    (64, 7, 9, 37),      # frames much smaller than a tile: every tile spans many frames
    (128, 28, 28, 5),    # 256-B LDS rows
    (128, 14, 14, 13),
])
@pytest.mark.parametrize("mode", [0, 1])
def test_conv3x3_staged_patch_bitwise(lib, c, H, W, n, mode):
    """The 3x3 kernel that stages the input patch once per 256 output pixels (conv3x3_halo_kernel):
    bit-identical to the oracle in BOTH math modes (bf16-MFMA model / f32 chain), including the
    frame borders, where a tap's LDS row belongs to a neighbouring image row or frame."""
    rng = np.random.default_rng(c + H * W + n)
    x = O.bf16_round((rng.standard_normal((n, H, W, c)) * np.exp2(rng.integers(-2, 3, (n, H, W, c)))).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((c, 3, 3, c)) * np.sqrt(2.0 / (9 * c))).astype(np.float32))
    b = (rng.standard_normal(c) * 0.2).astype(np.float32)
    acc = O.conv_acc_exact(x, w, 3, 3, 1, 1, mode="mfma" if mode == 0 else "f32")
    for relu in (1, 0):
        got = run_conv(lib, x, w, b, None, 1, 1, relu=relu, math_mode=mode)
        exp = O.epilogue(acc, b, res=None, relu=bool(relu))
        assert np.array_equal(got, exp), f"{np.mean(got != exp):.5f} of elements differ"


@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W,n,use_res", [
    (64, 64, 1, 1, 0, 14, 14, 3, False), (64, 128, 3, 2, 1, 15, 15, 2, True), (128, 256, 3, 1, 1, 9, 9, 2, False),
    (512, 128, 1, 1, 0, 7, 7, 3, True), (192, 64, 1, 1, 0, 16, 16, 2, False),
    (128, 256, 3, 1, 1, 30, 30, 19, False),   # 256x256 tile, 64-deep steps
    (512, 256, 1, 1, 0, 30, 30, 19, False),   # 256x256 tile on a 1x1
    (64, 64, 3, 1, 1, 56, 56, 2, True),       # 128x64 tile
])
def test_conv_production_mode_bitwise(lib, cin, cout, k, stride, pad, H, W, n, use_res):
    """The PRODUCTION kernels (bf16 MFMA) against the bit-exact model of the instruction
    (oracle/fav_exact.c: mfma_step8): same bits, for every tile shape and K depth."""
    rng = np.random.default_rng(cin * 7 + cout + k)
    x = O.bf16_round(np.maximum(rng.standard_normal((n, H, W, cin)) * np.exp2(rng.integers(-2, 3, (n, H, W, cin))), -0.3).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((cout, k, k, cin)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32))
    b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
    acc = O.conv_acc_exact(x, w, k, k, stride, pad, mode="mfma")
    got = run_conv(lib, x, w, b, None, stride, pad, relu=0, out_f32=1, math_mode=0)
    assert np.array_equal(got, acc + b), f"{np.mean(got != acc + b):.5f} of elements differ"
    res = O.bf16_round(rng.standard_normal(acc.shape).astype(np.float32)) if use_res else None
    got2 = run_conv(lib, x, w, b, res, stride, pad, relu=1, math_mode=0)
    assert np.array_equal(got2, O.epilogue(acc, b, res=res, relu=True))


def _exact_case(arch, blob, frames, first_index=0, hw=None, math="f32_exact", **kw):
    model = O.parse_blob(blob)
    aid = weights.ARCH_IDS[arch]
    policy = kw.get("dropout_policy", "none")
    T = kw.get("n_samples", 1)
    be = Backend(arch, blob, max_batch=frames.shape[0], math_mode=math, in_hw=hw, **kw)
    labels, conf = be.classify(torch.from_numpy(frames).cuda(), first_index=first_index)
    lg = be.logits().cpu().numpy()
    ocfg = O.ClassifyConfig(n_samples=T, site_mask=weights.site_mask_for(aid, policy), p=kw.get("dropout_p", 0.0),
                            seed=kw.get("seed", 0), exact=True if math == "f32_exact" else "mfma",
                            conf_kind=O.CONF_ENTROPY if kw.get("conf_kind") == "entropy" else O.CONF_MAX_SOFTMAX)
    ids = np.arange(first_index, first_index + frames.shape[0])
    ol, oc, olg, opb = O.classify(model, frames, ocfg, img_ids=ids, return_logits=True)
    be.close()
    assert lg.shape == olg.shape
    assert np.array_equal(lg, olg), f"logits differ: {np.mean(lg != olg):.4f} of elements, max {np.abs(lg - olg).max()}"
    srt = np.sort(opb, axis=1)
    tie = (srt[:, -1] - srt[:, -2]) < 1e-6   # the head's expf may differ by an ulp from NumPy's
    assert np.array_equal(labels.cpu().numpy()[~tie], ol[~tie])
    np.testing.assert_allclose(conf.cpu().numpy(), oc, rtol=0, atol=3e-6)


def test_resnet18_exact_single_pass(r18_blob):
    _exact_case("resnet18_cifar", r18_blob[0], synth.synthetic_frames_u8(32, 32, 32, seed=7))


@pytest.mark.parametrize("policy", ["last_layer", "layer4+fc", "all_blocks"])
def test_resnet18_exact_mc_dropout(r18_blob, policy):
    _exact_case("resnet18_cifar", r18_blob[0], synth.synthetic_frames_u8(10, 32, 32, seed=8), first_index=123,
                n_samples=4, dropout_policy=policy, dropout_p=0.1, seed=4, conf_kind="entropy", chunk_a=7, chunk_b=9)


def test_resnet50_exact_224(r50_blob):
    frames = synth.gaussian_noise_f32(synth.synthetic_frames_u8(3, 224, 224, seed=7), 3, seed=3)
    _exact_case("resnet50", r50_blob[0], frames)


def test_resnet50_exact_mc_dropout_small(r50_blob):
    frames = synth.synthetic_frames_u8(4, 96, 96, seed=9)
    _exact_case("resnet50", r50_blob[0], frames, first_index=5, hw=(96, 96), n_samples=3, dropout_policy="all_blocks",
                dropout_p=0.1, seed=4, chunk_a=3, chunk_b=5, regroup_block=9)
    _exact_case("resnet50", r50_blob[0], frames, hw=(96, 96), n_samples=2, dropout_policy="layer4+fc", dropout_p=0.2, seed=11)


def test_deep_ensemble_exact(r18_blob):
    """BASELINE configs[3] mechanics: M independently seeded members, softmax averaged over
    members by the same head that averages MC samples; logits [M][n][C] bit-exact per member."""
    blobs = [r18_blob[0]] + [weights.make_synthetic("resnet18_cifar", seed=s)[0] for s in (2, 3)]
    frames = synth.synthetic_frames_u8(16, 32, 32, seed=12)
    be = Backend("resnet18_cifar", blobs, max_batch=16, math_mode="f32_exact")
    labels, conf = be.classify(torch.from_numpy(frames).cuda())
    lg = be.logits().cpu().numpy()
    be.close()
    assert lg.shape == (3, 16, 10)
    ref = np.stack([O.classify(O.parse_blob(b), frames, O.ClassifyConfig(exact=True), return_logits=True)[2][0] for b in blobs])
    assert np.array_equal(lg, ref)
    ol, oc, _ = O.confidence_head(ref)
    assert np.array_equal(labels.cpu().numpy(), ol)
    np.testing.assert_allclose(conf.cpu().numpy(), oc, rtol=0, atol=3e-6)


# ---- the production mode itself, bit for bit ----------------------------------------------

def test_resnet18_production_mode_bitwise(r18_blob):
    """FAV_MATH_BF16 end to end against the oracle running the bit-exact MFMA model: logits
    bit-identical, labels exactly equal — no tolerance."""
    _exact_case("resnet18_cifar", r18_blob[0], synth.synthetic_frames_u8(16, 32, 32, seed=7), math="bf16")


@pytest.mark.parametrize("policy", ["last_layer", "layer4+fc", "all_blocks"])
def test_resnet18_production_mode_mc_dropout_bitwise(r18_blob, policy):
    _exact_case("resnet18_cifar", r18_blob[0], synth.synthetic_frames_u8(6, 32, 32, seed=8), first_index=123, math="bf16",
                n_samples=3, dropout_policy=policy, dropout_p=0.1, seed=4, conf_kind="entropy", chunk_a=5, chunk_b=7)


def test_resnet50_production_mode_bitwise(r50_blob):
    frames = synth.gaussian_noise_f32(synth.synthetic_frames_u8(2, 224, 224, seed=7), 3, seed=3)
    _exact_case("resnet50", r50_blob[0], frames, math="bf16")
    small = synth.synthetic_frames_u8(3, 64, 64, seed=9)
    _exact_case("resnet50", r50_blob[0], small, first_index=5, hw=(64, 64), math="bf16", n_samples=2,
                dropout_policy="all_blocks", dropout_p=0.1, seed=4)


def test_conv3x3_too_wide_for_the_staged_patch_falls_back_to_the_generic_kernel(lib):
    """A frame so wide that the staged-patch 3x3 kernel's LDS image (256 + 2W + 2 pixels) exceeds 160 KB: launch_conv
    must take the generic implicit-GEMM kernel and still match the oracle bit for bit."""
    rng = np.random.default_rng(77)
    n, H, W, c = 2, 4, 700, 64          # patch would be (256 + 1402) * 128 B = 212 KB; M = 5600 >= the 2048-row threshold
    x = O.bf16_round((rng.standard_normal((n, H, W, c)) * np.exp2(rng.integers(-2, 3, (n, H, W, c)))).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((c, 3, 3, c)) * np.sqrt(2.0 / (9 * c))).astype(np.float32))
    b = (rng.standard_normal(c) * 0.2).astype(np.float32)
    got = run_conv(lib, x, w, b, None, 1, 1, relu=1, math_mode=0)
    ref = O.epilogue(O.conv_acc_exact(x, w, 3, 3, 1, 1, mode="mfma"), b, relu=True)
    assert np.array_equal(got, ref), f"{np.mean(got != ref):.5f} of elements differ"


def test_resnet50_wide_frames_mix_fused_tails_and_fallbacks(r50_blob):
    """64 x 1408 frames: layer 1's fused tails still fit their LDS image (W = 352), layer 2's (W = 176, 256-pixel tiles)
    do not and fall back to the separate launches - the schedule mixes both and the logits stay bit-identical to the
    MFMA-model oracle."""
    blob, _ = r50_blob
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(2, 64, 1408, seed=13)
    be = Backend("resnet50", blob, in_hw=(64, 1408), max_batch=2, n_samples=2, dropout_policy="all_blocks", dropout_p=0.1, seed=4)
    be.classify(torch.from_numpy(frames).cuda(), first_index=5)
    got = be.logits().cpu().numpy()
    be.close()
    cfg = O.ClassifyConfig(n_samples=2, site_mask=weights.site_mask_for(1, "all_blocks"), p=0.1, seed=4, exact="mfma")
    ref = O.classify(model, frames, cfg, img_ids=np.arange(5, 7), return_logits=True)[2]
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("n,H,W", [(2, 224, 224), (3, 64, 80), (1, 33, 47), (1, 240, 320), (5, 9, 7)])
def test_stem_pool_fused_bitwise(lib, layout, n, H, W):
    """The ImageNet stem as ONE launch (stem7_pool_kernel: normalise, 7x7/2 convolution gathered from an LDS patch, bias,
    ReLU, 3x3/2 max pool) against the oracle's normalise -> im2col -> MFMA-model GEMM -> epilogue -> max pool: same
    bits for both frame layouts, frame sizes that are no multiple of the 8 x 8 pool tile, and frames smaller than a tile."""
    import ctypes as C
    rng = np.random.default_rng(H * 1000 + W + n)
    img = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8) if layout == 0 else rng.random((n, H, W, 3), dtype=np.float32)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    istd = O.inv_std32(std)
    w = O.bf16_round((rng.standard_normal((64, 7, 7, 3)) * np.sqrt(2.0 / 147)).astype(np.float32))
    b = (rng.standard_normal(64) * 0.2).astype(np.float32)
    cols, ho, wo = O._im2col(O.normalize_input(img, mean, istd), 7, 7, 2, 3)
    a = np.zeros((n, ho, wo, 192), np.float32)
    a[..., :147] = cols.reshape(n, ho, wo, 147)
    wp = np.zeros((64, 1, 1, 192), np.float32)
    wp[:, 0, 0, :147] = w.reshape(64, 147)
    ref = O.maxpool3x3s2(O.epilogue(O.conv_acc_exact(a, wp, 1, 1, 1, 0, mode="mfma"), b, res=None, relu=True))
    out = torch.empty(ref.shape, dtype=torch.bfloat16, device="cuda")
    m3 = (C.c_float * 3)(*mean)
    i3 = (C.c_float * 3)(*[float(v) for v in istd])
    wd = torch.from_numpy(wp.reshape(64, 192)).cuda().to(torch.bfloat16).contiguous()
    bd = torch.from_numpy(b).cuda()
    _lib.check(lib.fav_op_stem_pool(torch.from_numpy(img).cuda().data_ptr(), layout, n, H, W, wd.data_ptr(), bd.data_ptr(), m3, i3,
                                    out.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.to(torch.float32).cpu().numpy()
    assert np.array_equal(got, ref), f"{np.mean(got != ref):.5f} of elements differ"
