"""CPU checks of the ViT oracle (BASELINE configs[4]).  The oracle is the specification (PARITY UNPINNED:
the reference has no classifier), so it is cross-checked against torch on CPU - exp, GELU, LayerNorm and
softmax against float64 formulas, the whole encoder against a torch.nn.functional implementation - and the
synthetic checkpoints are pinned by checksum."""
import json
import os

import numpy as np
import pytest

from failure_aware_vision_amd import synth, weights
from oracle import fav_oracle as O

torch = pytest.importorskip("torch")
F = torch.nn.functional


def test_exact_exp_and_gelu_are_accurate():
    x = np.concatenate([np.linspace(-80, 88, 200001), np.linspace(-1, 1, 100001)]).astype(np.float32)
    e = O.expf_exact(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 4e-7                      # ~3 ulp of fp32
    assert O.expf_exact(np.array([-81.0, -1e30, -np.inf], np.float32)).tolist() == [0.0, 0.0, 0.0]
    xg = np.linspace(-12, 12, 240001).astype(np.float32)
    g = O.gelu_exact(xg).astype(np.float64)
    refg = F.gelu(torch.from_numpy(xg).double()).numpy()              # the erf form, torch.nn.GELU's default
    assert np.max(np.abs(g - refg)) < 1e-4                           # (the tanh form is 4.7e-4 away from it)
    assert O.gelu_exact(np.array([-30.0, 30.0], np.float32)).tolist() == [-0.0, 30.0]


def test_layernorm_and_softmax_vs_torch():
    rng = np.random.default_rng(0)
    for d in (128, 768, 1024):
        x = (rng.standard_normal((9, d)) * 3 + 1).astype(np.float32)
        g, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
        ref = F.layer_norm(torch.from_numpy(x).double(), (d,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), 1e-6)
        assert np.abs(O.layernorm_exact(x, g, b) - ref.numpy()).max() < 2e-5
    s = (rng.standard_normal((11, 197)) * 4).astype(np.float32)
    p = O.attn_softmax_exact(s)                                      # raw scores in, softmax(s / 8) out
    assert np.abs(p - torch.softmax(torch.from_numpy(s).double() / 8, -1).numpy()).max() < 3e-6


def torch_vit_logits(model, xn):
    """fp64 torch restatement of the same encoder (no bf16 rounding): what the oracle approximates."""
    c, Ls = O.VIT_CFG[model.arch], model.layers
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).double()
    b, hh, ww, _ = xn.shape
    p, d, heads = c["patch"], c["dim"], c["heads"]
    x = F.conv2d(t(xn).permute(0, 3, 1, 2), t(Ls[0].w).permute(0, 3, 1, 2), t(Ls[0].b), stride=p)    # [b, d, gh, gw]
    x = x.flatten(2).transpose(1, 2)
    pos = t(Ls[1].w).reshape(-1, d)
    x = torch.cat([pos[:1].expand(b, 1, d), x + pos[1:]], 1)
    li = 2
    for _ in range(c["depth"]):
        ln1, qkv, proj, ln2, fc1, fc2 = Ls[li:li + 6]
        li += 6
        y = F.layer_norm(x, (d,), t(ln1.w), t(ln1.b), 1e-6)
        q, k, v = F.linear(y, t(qkv.w).reshape(3 * d, d), t(qkv.b)).reshape(b, -1, 3, heads, 64).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, -1, d)
        x = x + F.linear(a, t(proj.w).reshape(d, d), t(proj.b))
        y = F.layer_norm(x, (d,), t(ln2.w), t(ln2.b), 1e-6)
        x = x + F.linear(F.gelu(F.linear(y, t(fc1.w).reshape(-1, d), t(fc1.b))), t(fc2.w).reshape(d, -1), t(fc2.b))
    y = F.layer_norm(x[:, 0], (d,), t(Ls[li].w), t(Ls[li].b), 1e-6)
    return F.linear(y, t(Ls[li + 1].w).reshape(-1, d), t(Ls[li + 1].b)).numpy()


@pytest.mark.parametrize("exact", [False, True, "mfma"])
def test_vit_tiny_oracle_vs_torch(exact):
    blob, info = weights.make_synthetic_vit("vit_tiny", seed=3)
    model = O.parse_blob(blob)
    frames = synth.synthetic_frames_u8(5, 64, 64, seed=11)
    cfg = O.ClassifyConfig(exact=exact, temperature=1.5, conf_kind=O.CONF_ENTROPY)
    xn = O.normalize_input(frames, cfg.mean, O.inv_std32(cfg.std))
    labels, conf, lg, pbar = O.classify(model, frames, cfg, return_logits=True)
    ref = torch_vit_logits(model, xn)
    # bf16 activations vs an fp64 network: small relative to the logits' spread, same winners where the gap is clear
    assert np.sqrt(np.mean((lg[0] - ref) ** 2)) < 0.03 * ref.std()
    pr = torch.softmax(torch.from_numpy(ref) / 1.5, -1).numpy()
    srt = np.sort(pr, 1)
    clear = (srt[:, -1] - srt[:, -2]) > 0.05
    assert np.array_equal(labels[clear], pr.argmax(1)[clear])
    ent = 1.0 + (pr * np.log(np.maximum(pr, 1e-300))).sum(1) / np.log(pr.shape[1])
    assert np.abs(conf - ent).max() < 0.02


def test_vit_checkpoints_are_machine_independent():
    path = os.path.join(os.path.dirname(__file__), "golden", "blob_sha256.json")
    pins = json.load(open(path))
    assert weights.make_synthetic_vit("vit_tiny", seed=1)[1]["sha256"] == pins["vit_tiny_seed1"]
    assert weights.make_synthetic_vit("vit_b16", seed=1)[1]["sha256"] == pins["vit_b16_seed1"]


def test_attention_key_order_is_a_block_permutation():
    """Slot order of the second attention product: a permutation inside every 32-key block, identity on the block index,
    keys 4g .. 4g + 3 and 16 + 4g .. 16 + 4g + 3 in lane group g's eight slots."""
    for t in (1, 17, 32, 197, 256):
        order = O.attn_key_order(t)
        assert order.size == (t + 31) // 32 * 32
        assert np.array_equal(np.sort(order), np.arange(order.size))
        assert np.array_equal(order >> 5, np.arange(order.size) >> 5)
    assert O.attn_key_order(32)[:16].tolist() == [0, 1, 2, 3, 16, 17, 18, 19, 4, 5, 6, 7, 20, 21, 22, 23]


def test_attention_does_not_depend_on_how_keys_are_padded():
    """attention() pads keys to a multiple of 32 with zero probabilities: two token counts with the same real keys and queries
    give the same rows (what lets the device mask only the last key tile)."""
    rng = np.random.default_rng(4)
    qkv = O.bf16_round((rng.standard_normal((1, 40, 3 * 64)) * 1.1).astype(np.float32))
    full = O.attention(qkv, 1, exact="mfma")
    again = O.attention(qkv.copy(), 1, exact="mfma")
    assert np.array_equal(full, again)
    ref = np.einsum("qk,kd->qd", torch.softmax(torch.from_numpy(qkv[0, :, :64] @ qkv[0, :, 64:128].T).double() / 8, -1).numpy(),
                    qkv[0, :, 128:].astype(np.float64))
    assert np.abs(full[0] - ref).max() < 0.03
