"""ViT-B/16 path (BASELINE configs[4]): attention (MFMA QK^T / softmax / V), LayerNorm, GELU MLP,
temperature-scaled entropy confidence.

exp, GELU, LayerNorm and the attention softmax are fixed sequences of IEEE fp32 operations that
oracle/fav_exact.c restates, and the matrix products go through the same summation models as the
convolutions (bit-exact model of v_mfma_f32_16x16x32_bf16 in production mode, k-ordered fp32 chain in
validation mode), so every kernel and the whole encoder must agree with the oracle BIT FOR BIT."""
import ctypes as C

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


def dev_bf16(x):
    """fp32 array holding bf16 values -> torch bf16 CUDA tensor (exact)."""
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda().to(torch.bfloat16)


def host_f32(t):
    return t.float().cpu().numpy()


@pytest.mark.parametrize("rows,D,stride_mul", [(20000, 768, 1), (4099, 768, 2), (4097, 192, 1), (5, 128, 1), (6, 768, 3), (3, 1024, 1)])
def test_layernorm_bitwise(lib, rows, D, stride_mul):
    rng = np.random.default_rng(rows + D)
    # many rows with different scales: a square root or division that is not correctly rounded shows up as a
    # one-ulp rstd in ~10 % of the rows and flips a bf16 rounding in about one row in a hundred
    x = O.bf16_round((rng.standard_normal((rows, stride_mul * D)) * np.exp2(rng.integers(-2, 3, (rows, 1))) * 2.5
                      + rng.standard_normal((rows, 1))).astype(np.float32))
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.05 * rng.standard_normal(D)).astype(np.float32)
    xd, gd, bd = dev_bf16(x), torch.from_numpy(g).cuda(), torch.from_numpy(b).cuda()
    y = torch.empty((rows, D), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_layernorm(xd.data_ptr(), stride_mul * D, gd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, D,
                                    C.c_float(1e-6), None))
    torch.cuda.synchronize()
    exp = O.bf16_round(O.layernorm_exact(x[:, :D], g, b))
    assert np.array_equal(host_f32(y), exp)


@pytest.mark.parametrize("n,T,heads", [(2, 197, 12), (3, 17, 2), (1, 256, 1), (2, 33, 3), (1, 208, 2), (2, 193, 1), (1, 1, 1), (1, 130, 2)])
@pytest.mark.parametrize("mode", [0, 1])
def test_attention_bitwise(lib, n, T, heads, mode):
    rng = np.random.default_rng(n * 1000 + T + heads)
    D = heads * 64
    qkv = O.bf16_round((rng.standard_normal((n, T, 3 * D)) * 1.2).astype(np.float32))
    qd = dev_bf16(qkv)
    out = torch.empty((n, T, D), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fav_op_attention(qd.data_ptr(), out.data_ptr(), n, T, D, heads, mode, None))
    torch.cuda.synchronize()
    exp = O.attention(qkv, heads, exact="mfma" if mode == 0 else True)
    got = host_f32(out)
    assert np.array_equal(got, exp), f"{np.mean(got != exp):.5f} of elements differ, max {np.abs(got - exp).max()}"
    # and it is attention: against a float64 reference within bf16 resolution
    q, k, v = (qkv[..., i * D:(i + 1) * D].reshape(n, T, heads, 64).astype(np.float64) for i in range(3))
    s = np.einsum("bqhd,bkhd->bhqk", q, k) / 8.0
    p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bkhd->bqhd", p, v).reshape(n, T, D)
    assert np.abs(got - ref).max() < 0.03


@pytest.mark.parametrize("mode", [0, 1])
def test_linear_gelu_epilogue_bitwise(lib, mode):
    rng = np.random.default_rng(5)
    x = O.bf16_round(rng.standard_normal((3, 19, 1, 128)).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((256, 1, 1, 128)) / np.sqrt(128) * 2).astype(np.float32))
    b = (rng.standard_normal(256) * 0.3).astype(np.float32)
    acc = O.conv_acc_exact(x, w, 1, 1, 1, 0, mode="mfma" if mode == 0 else True)
    got = run_conv(lib, x, w, b, None, 1, 0, relu=2, math_mode=mode)
    assert np.array_equal(got, O.bf16_round(O.gelu_exact(acc + b)))


def test_vit_assemble_bitwise(lib):
    rng = np.random.default_rng(9)
    n, ntok, D = 3, 17, 128
    emb = O.bf16_round(rng.standard_normal((n, ntok - 1, D)).astype(np.float32))
    pos = (0.2 * rng.standard_normal((ntok, D))).astype(np.float32)
    x = torch.empty((n, ntok, D), dtype=torch.bfloat16, device="cuda")
    ed, pd = dev_bf16(emb), torch.from_numpy(pos).cuda()
    _lib.check(lib.fav_op_vit_assemble(ed.data_ptr(), pd.data_ptr(), x.data_ptr(), n, ntok, D, None))
    torch.cuda.synchronize()
    exp = np.empty((n, ntok, D), np.float32)
    exp[:, 0] = pos[0]
    exp[:, 1:] = emb + pos[1:]
    assert np.array_equal(host_f32(x), O.bf16_round(exp))


def _vit_case(arch, frames, math, **kw):
    blob, info = weights.make_synthetic_vit(arch, seed=3, in_hw=frames.shape[1:3])
    model = O.parse_blob(blob)
    be = Backend(arch, blob, max_batch=frames.shape[0], in_hw=frames.shape[1:3], math_mode=math, temperature=1.5,
                 conf_kind="entropy", **kw)
    labels, conf = be.classify(torch.from_numpy(frames).cuda())
    lg = be.logits().cpu().numpy()
    be.close()
    ocfg = O.ClassifyConfig(exact=True if math == "f32_exact" else "mfma", temperature=1.5, conf_kind=O.CONF_ENTROPY)
    ol, oc, olg, opb = O.classify(model, frames, ocfg, return_logits=True)
    assert lg.shape == olg.shape
    assert np.array_equal(lg, olg), f"logits differ: {np.mean(lg != olg):.4f} of elements, max {np.abs(lg - olg).max()}"
    srt = np.sort(opb, axis=1)
    tie = (srt[:, -1] - srt[:, -2]) < 1e-6
    assert np.array_equal(labels.cpu().numpy()[~tie], ol[~tie])
    np.testing.assert_allclose(conf.cpu().numpy(), oc, rtol=0, atol=3e-6)
    return ol


@pytest.mark.parametrize("math", ["bf16", "f32_exact"])
@pytest.mark.parametrize("hw", [(64, 64), (48, 80)])
def test_vit_tiny_bitwise(math, hw):
    """Two encoder layers, 128 wide, 2 heads: every logit bit-identical to the oracle in both math modes."""
    _vit_case("vit_tiny", synth.synthetic_frames_u8(6, hw[0], hw[1], seed=11), math)


def test_vit_b16_production_mode_bitwise():
    """The full ViT-B/16 (12 layers, 768 wide, 12 heads, 197 tokens) at 224x224, production bf16 mode, corrupted
    frames: all 1000 logits of every frame bit-identical to the oracle; entropy confidence within 3e-6."""
    u8 = synth.synthetic_frames_u8(2, 224, 224, seed=21)
    frames = synth.gaussian_noise_f32(u8, 3, seed=3)
    _vit_case("vit_b16", frames, "bf16")


def test_vit_rejects_what_it_does_not_support():
    blob, _ = weights.make_synthetic_vit("vit_tiny", seed=3)
    with pytest.raises(_lib.FavError):
        Backend("vit_tiny", blob, in_hw=(72, 64))                     # not a multiple of the patch
    with pytest.raises(_lib.FavError):
        Backend("vit_tiny", blob, in_hw=(64, 64), site_mask=1, dropout_p=0.1, n_samples=4)   # no dropout sites
    with pytest.raises(_lib.FavError):
        Backend("vit_tiny", blob, in_hw=(128, 128))                   # position table of another grid
