"""Synthetic checkpoints and the "FAVW" weight blob that fav_load_weights() takes.

No pretrained weights exist offline and the reference ships none (SURVEY.md
§8c "Remote loaders"), so checkpoints are generated: He-normal convolutions with
BatchNorm statistics *calibrated* on a small batch of synthetic frames (as a
trained network's running statistics would be), then folded into per-output-
channel scale (multiplied into the weights before the bf16 rounding) and an
fp32 bias.  Calibrated statistics are quantised to bf16 before folding so the
resulting blob is bit-identical on any machine (BLAS summation order only moves
the statistics by ~1e-16 relative); tests pin the blob by checksum.

Blob layout (little endian):
  header  32 B : u32 magic 'FAVW', u32 version=1, u32 arch, u32 num_classes,
                 u32 n_layers, u32 reserved[3]
  table   48 B per layer: u32 cout, cin, kh, kw, stride, pad, reserved[2];
                 u64 w_off, b_off   (byte offsets from blob start, 64-B aligned)
  data         : per layer bf16 w[cout][kh][kw][cin] (BN scale folded in),
                 fp32 b[cout]
Layer order: stem; per block conv1, conv2, (conv3), (downsample); fc.

ViT checkpoints (arch "vit_b16", "vit_tiny") use the same container.  A table row with kh = 0
is a pair of fp32 vectors of length cout (w_off -> first, b_off -> second): LayerNorm
(gamma, beta), and the position table (pos[ntok][D] with the class token added into row 0;
second vector unused).  Order: patch embedding (cout = D, cin = 3, kh = kw = stride = patch);
position table; per block: ln1, qkv [3D][D] (rows Q | K | V, head h = columns 64h..64h+63 of
each), proj, ln2, fc1, fc2; final ln; head.
"""
from __future__ import annotations

import hashlib
import struct

import numpy as np

from . import synth

BLOB_MAGIC = 0x57564146
ARCH_IDS = {"resnet18_cifar": 0, "resnet50": 1, "vit_b16": 2, "vit_tiny": 3}
VIT_CFG = {  # embed dim, depth, heads (64-wide), MLP width, patch, default input
    2: dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, in_hw=(224, 224)),
    3: dict(dim=128, depth=2, heads=2, mlp=256, patch=16, in_hw=(64, 64)),
}
_ARCH = {
    0: dict(block="basic", depths=(2, 2, 2, 2), planes=(64, 128, 256, 512), stem="cifar", calib_hw=32, n_calib=16),
    1: dict(block="bottleneck", depths=(3, 4, 6, 3), planes=(64, 128, 256, 512), stem="imagenet", calib_hw=224, n_calib=8),
}
DEFAULT_MEAN = (0.485, 0.456, 0.406)
DEFAULT_STD = (0.229, 0.224, 0.225)


def layer_specs(arch: int, num_classes: int):
    """Blob-order list of dict(cout, cin, kh, kw, stride, pad, role, block)."""
    a = _ARCH[arch]
    L = []
    if a["stem"] == "imagenet":
        L.append(dict(cout=64, cin=3, kh=7, kw=7, stride=2, pad=3, role="stem", block=-1))
    else:
        L.append(dict(cout=64, cin=3, kh=3, kw=3, stride=1, pad=1, role="stem", block=-1))
    exp = 4 if a["block"] == "bottleneck" else 1
    inpl, bidx = 64, 0
    for li, (d, p) in enumerate(zip(a["depths"], a["planes"])):
        for bi in range(d):
            s = 2 if (bi == 0 and li > 0) else 1
            if a["block"] == "bottleneck":
                L.append(dict(cout=p, cin=inpl, kh=1, kw=1, stride=1, pad=0, role="mid", block=bidx))
                L.append(dict(cout=p, cin=p, kh=3, kw=3, stride=s, pad=1, role="mid", block=bidx))
                L.append(dict(cout=p * 4, cin=p, kh=1, kw=1, stride=1, pad=0, role="last", block=bidx))
            else:
                L.append(dict(cout=p, cin=inpl, kh=3, kw=3, stride=s, pad=1, role="mid", block=bidx))
                L.append(dict(cout=p, cin=p, kh=3, kw=3, stride=1, pad=1, role="last", block=bidx))
            if bi == 0 and (s != 1 or inpl != p * exp):
                L.append(dict(cout=p * exp, cin=inpl, kh=1, kw=1, stride=s, pad=0, role="down", block=bidx))
            inpl = p * exp
            bidx += 1
    L.append(dict(cout=num_classes, cin=inpl, kh=1, kw=1, stride=1, pad=0, role="fc", block=bidx))
    return L


def n_blocks(arch: int) -> int:
    if arch in VIT_CFG:
        return 0
    return sum(_ARCH[arch]["depths"])


def site_mask_for(arch: int, policy: str) -> int:
    """Dropout-site bitmask.  Site s < n_blocks is the output of residual block
    s; site n_blocks is the pooled feature vector feeding the classifier."""
    if policy in ("none", "", None):
        return 0
    if arch in VIT_CFG:
        raise ValueError("the ViT path has no dropout sites (BASELINE configs[4] is a single pass)")
    nb = n_blocks(arch)
    if policy == "last_layer":
        return 1 << nb
    if policy == "all_blocks":
        return (1 << nb) - 1
    if policy == "layer4+fc":
        d = _ARCH[arch]["depths"]
        first = nb - d[-1] - 1  # output of the block feeding the last stage
        m = 1 << nb
        for s in range(first, nb - 1):
            m |= 1 << s
        return m
    raise ValueError(f"unknown dropout policy {policy!r}")


def _bf16(x):
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _bf16_f32(x):
    return (_bf16(x).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(x))


def _conv64(x, w, stride, pad):
    """float64 NHWC conv by im2col (calibration only)."""
    b, h, ww, c = x.shape
    co, kh, kw, _ = w.shape
    if kh == 1 and kw == 1 and pad == 0:
        xs = x[:, ::stride, ::stride]
        return (xs.reshape(-1, c) @ w.reshape(co, c).T.astype(np.float64)).reshape(xs.shape[:3] + (co,))
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (ww + 2 * pad - kw) // stride + 1
    xp = np.zeros((b, h + 2 * pad, ww + 2 * pad, c))
    xp[:, pad:pad + h, pad:pad + ww] = x
    sb, sh, sw, sc = xp.strides
    cols = np.lib.stride_tricks.as_strided(
        xp, (b, ho, wo, kh, kw, c), (sb, sh * stride, sw * stride, sh, sw, sc), writeable=False)
    return (cols.reshape(b * ho * wo, -1) @ w.reshape(co, -1).T.astype(np.float64)).reshape(b, ho, wo, co)


def make_synthetic_vit(arch="vit_b16", seed: int = 1, num_classes: int = 1000, in_hw=None,
                       mean=DEFAULT_MEAN, std=DEFAULT_STD, logit_std: float = 5.0):
    """Seeded synthetic ViT checkpoint (pre-norm encoder, class token, learned positions,
    GELU MLP).  LayerNorm keeps activations in range, so no calibration pass is needed:
    linear layers are N(0, 1/fan_in) (x0.5 where they feed the residual stream) and the head is
    scaled for logits of standard deviation `logit_std`.  Returns (blob, info)."""
    aid = ARCH_IDS[arch] if isinstance(arch, str) else int(arch)
    cfg = VIT_CFG[aid]
    D, depth, mlp, P = cfg["dim"], cfg["depth"], cfg["mlp"], cfg["patch"]
    H, W = in_hw or cfg["in_hw"]
    if H % P or W % P:
        raise ValueError("ViT input must be a multiple of the patch size")
    ntok = (H // P) * (W // P) + 1
    rng = np.random.default_rng(seed)
    specs, folded = [], []

    def linear(cout, cin, gain, kh=1, kw=1, stride=1):
        w = (rng.standard_normal((cout, kh, kw, cin)) * (gain / np.sqrt(kh * kw * cin))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.02).astype(np.float32)
        specs.append(dict(cout=cout, cin=cin, kh=kh, kw=kw, stride=stride, pad=0))
        folded.append((_bf16(w), b))

    def vectors(a, b):
        specs.append(dict(cout=a.size, cin=0, kh=0, kw=0, stride=0, pad=0))
        folded.append((np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)))

    def layernorm():
        vectors(1.0 + 0.1 * rng.standard_normal(D), 0.05 * rng.standard_normal(D))

    linear(D, 3, 1.0, P, P, P)
    pos = (0.2 * rng.standard_normal((ntok, D))).astype(np.float32)
    pos[0] += (0.2 * rng.standard_normal(D)).astype(np.float32)   # class token
    vectors(pos.reshape(-1), np.zeros(ntok * D, np.float32))
    for _ in range(depth):
        layernorm(); linear(3 * D, D, 1.0); linear(D, D, 0.5)
        layernorm(); linear(mlp, D, 1.0); linear(D, mlp, 0.5)
    layernorm()
    linear(num_classes, D, logit_std)
    folded[-1] = (folded[-1][0], np.zeros(num_classes, np.float32))
    blob = pack_blob(aid, num_classes, specs, folded)
    info = dict(arch=arch, num_classes=num_classes, n_layers=len(specs), seed=seed, in_hw=(H, W), n_tokens=ntok,
                sha256=hashlib.sha256(blob).hexdigest(), mean=tuple(mean), std=tuple(std))
    return blob, info


def tv_names(arch: int, specs):
    """torchvision state_dict prefixes for the blob-order layer list -> [(conv_name, bn_name | None)]."""
    a = _ARCH[arch]
    names, seen = [], {}
    for sp in specs:
        if sp["role"] == "stem":
            names.append(("conv1", "bn1"))
        elif sp["role"] == "fc":
            names.append(("fc", None))
        else:
            blk, stage = sp["block"], 0
            while blk >= a["depths"][stage]:
                blk -= a["depths"][stage]; stage += 1
            p = f"layer{stage + 1}.{blk}."
            if sp["role"] == "down":
                names.append((p + "downsample.0", p + "downsample.1"))
            else:
                k = seen[p] = seen.get(p, 0) + 1
                names.append((f"{p}conv{k}", f"{p}bn{k}"))
    return names


def _fold(w_ohwi, gamma, beta, mean, var, eps):
    """fp32 parameters -> (bf16 weight bits [O,kh,kw,I], fp32 bias): the fold `from_state_dict` applies."""
    w, gamma, beta, mean, var = (np.asarray(t, np.float64) for t in (w_ohwi, gamma, beta, mean, var))
    scale = gamma / np.sqrt(var + eps)
    return _bf16((w * scale[:, None, None, None]).astype(np.float32)), (beta + (0.0 - mean) * scale).astype(np.float32)


def _head_from_features(rng, feat_det, feat_mc, num_classes, logit_std, k, gamma=0.5):
    """The classifier of a synthetic checkpoint -> (W fp32 [classes, C], bias-centre fp64 [C]).

    An untrained network's pooled features move far more with the *regime* (deterministic pass vs a dropout
    sample: the shift q1 is longer than the feature vector itself) and with the dropout noise (white, ~2x the
    between-frame spread in norm) than with the frame's content, so a dense random head labels frames by
    regime and noise.  A trained head reads the directions that carry content; this one does too: the top-k
    principal directions of the deterministic calibration features (deflated by q1, partially whitened),
    mixed into `num_classes` logits by a seeded Gaussian matrix."""
    m_det, m_mc = feat_det.mean(axis=0), feat_mc.mean(axis=0)
    q1 = m_mc - m_det
    q1 = q1 / max(np.linalg.norm(q1), 1e-30)
    x = feat_det - m_det
    x = x - np.outer(x @ q1, q1)
    _, s, vt = np.linalg.svd(x, full_matrices=False)
    k = int(min(k, max(1, len(x) - 2)))
    pcs = vt[:k].copy()
    for i in range(k):                                    # fix the sign (an SVD leaves it open)
        if pcs[i, np.argmax(np.abs(pcs[i]))] < 0:
            pcs[i] = -pcs[i]
    sig = np.maximum(s[:k] / np.sqrt(len(x)), 1e-12)
    mix = rng.standard_normal((num_classes, k))
    w = (mix / sig ** gamma) @ pcs
    centre = 0.5 * (m_det + m_mc)
    raw = np.concatenate([feat_det - centre, feat_mc - centre]) @ w.T
    g = float(_bf16_f32(np.float32(logit_std / max(raw.std(), 1e-12))))
    return (w * g).astype(np.float32), centre


def make_synthetic_state_dict(arch="resnet50", seed: int = 1, num_classes: int | None = None,
                              mean=DEFAULT_MEAN, std=DEFAULT_STD, logit_std: float = 12.0, n_calib: int | None = None,
                              calib_dropout_p: float = 0.1, head_rank: int = 4):
    """The synthetic checkpoint as an fp32 ``state_dict`` in torchvision naming (numpy arrays: conv OIHW,
    BatchNorm weight / bias / running_mean / running_var, fc) -> (state_dict, meta).  This is the checkpoint:
    ``make_synthetic`` is its fold (``from_state_dict``), and ``tests/torch_models.ResNet`` loads it as is.

    The calibration frames (at the deployment resolution: half of them with Gaussian noise of severities 1..5)
    run through every layer twice - as they are and with Bernoulli(calib_dropout_p) dropout on every
    residual-block output - so the running statistics suit both the deterministic pass and the MC-Dropout
    samples, as those of a network trained with dropout would.  Statistics are quantised to bf16 so the
    checkpoint is bit-identical on any machine (BLAS summation order only moves them by ~1e-16 relative)."""
    arch = ARCH_IDS[arch] if isinstance(arch, str) else int(arch)
    a = _ARCH[arch]
    if num_classes is None:
        num_classes = 1000 if arch == 1 else 10
    if n_calib is None:
        n_calib = a["n_calib"]
    specs = layer_specs(arch, num_classes)
    names = tv_names(arch, specs)
    rng = np.random.default_rng([int(seed), arch, 0x78])
    hw = a["calib_hw"]
    frames = synth.synthetic_frames_u8(n_calib, hw, hw, seed=0xCA11B, start_id=0)
    x01 = frames.astype(np.float64) / 255.0
    for i in range(n_calib // 2, n_calib):
        x01[i] = synth.gaussian_noise_f32(frames[i:i + 1], 1 + i % 5, seed=0xCA11B, start_id=i)[0]
    x = (x01 - np.asarray(mean)) / np.asarray(std)
    x = np.concatenate([x, x], axis=0)          # second half: the dropout pass
    drop_rng = np.random.default_rng([int(seed), arch, 0xD0])
    eps = 1e-5
    sd = {}

    def make_layer(spec, name, inp):
        conv, bn = name
        fan_in = spec["kh"] * spec["kw"] * spec["cin"]
        gain = 1.0 if spec["role"] == "down" else 2.0
        w = rng.standard_normal((spec["cout"], spec["kh"], spec["kw"], spec["cin"])) * np.sqrt(gain / fan_in)
        if spec["role"] == "stem" and spec["kh"] >= 5:
            # smooth (low-pass) first-layer filters, as a trained stem has: white pixel noise then excites the
            # network far less than image content does
            k1 = np.array([1.0, 4.0, 6.0, 4.0, 1.0]) / 16.0
            for ax in (1, 2):
                w = np.apply_along_axis(lambda v: np.convolve(v, k1, mode="same"), ax, w)
        w32 = w.astype(np.float32)
        acc = _conv64(inp, w32.astype(np.float64), spec["stride"], spec["pad"])
        mu = _bf16_f32(acc.mean(axis=(0, 1, 2)))
        var = _bf16_f32(acc.var(axis=(0, 1, 2)))
        gamma = {"stem": 1.0, "mid": 1.0, "last": 0.25, "down": 0.7}[spec["role"]]
        gam = (gamma * (1.0 + 0.1 * rng.standard_normal(spec["cout"]))).astype(np.float32)
        beta = ((0.25 if spec["role"] in ("stem", "mid") else 0.1) * rng.standard_normal(spec["cout"])).astype(np.float32)
        sd[conv + ".weight"] = np.ascontiguousarray(w32.transpose(0, 3, 1, 2))
        sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"], sd[bn + ".running_var"] = gam, beta, mu, var
        wq, bq = _fold(w32, gam, beta, mu, var, eps)
        wf = (wq.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        return _conv64(inp, wf, spec["stride"], spec["pad"]) + bq.astype(np.float64)

    it = iter(zip(specs, names))
    spec, name = next(it)
    act = np.maximum(make_layer(spec, name, x), 0.0)
    if a["stem"] == "imagenet":  # 3x3/2 max pool, pad 1
        b_, h_, w_, c_ = act.shape
        ho, wo = (h_ - 1) // 2 + 1, (w_ - 1) // 2 + 1
        xp = np.full((b_, h_ + 2, w_ + 2, c_), -np.inf)
        xp[:, 1:1 + h_, 1:1 + w_] = act
        act = np.max(np.stack([xp[:, r:r + 2 * ho:2, s:s + 2 * wo:2] for r in range(3) for s in range(3)]), axis=0)
    nmain = 3 if a["block"] == "bottleneck" else 2
    spec, name = next(it)
    while spec["role"] != "fc":
        blk = spec["block"]
        h = act
        main = [(spec, name)] + [next(it) for _ in range(nmain - 1)]
        for sp, nm in main[:-1]:
            h = np.maximum(make_layer(sp, nm, h), 0.0)
        branch = make_layer(main[-1][0], main[-1][1], h)
        spec, name = next(it)
        idn = act
        if spec["role"] == "down" and spec["block"] == blk:
            idn = make_layer(spec, name, act)
            spec, name = next(it)
        act = np.maximum(branch + idn, 0.0)
        if calib_dropout_p > 0:
            keep = drop_rng.random(act[n_calib:].shape) >= calib_dropout_p
            act[n_calib:] = np.where(keep, act[n_calib:] / (1.0 - calib_dropout_p), 0.0)
    feat = act.mean(axis=(1, 2))  # [2*n_calib, C]
    w32, centre = _head_from_features(rng, feat[:n_calib], feat[n_calib:], num_classes, logit_std, head_rank)
    wq = _bf16(w32)
    wf = (wq.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    fcentre = _bf16_f32(centre).astype(np.float64)
    sd["fc.weight"] = w32
    sd["fc.bias"] = (-(fcentre @ wf.T)).astype(np.float32)     # centred on what the bf16 weights see
    meta = dict(arch=arch, num_classes=num_classes, seed=seed, bn_eps=eps, mean=tuple(mean), std=tuple(std),
                calib_hw=hw, n_calib=n_calib)
    return sd, meta


def make_synthetic(arch="resnet50", seed: int = 1, num_classes: int | None = None,
                   mean=DEFAULT_MEAN, std=DEFAULT_STD, **kw):
    """Returns (blob bytes, info dict): the fold of ``make_synthetic_state_dict`` (same arguments)."""
    sd, meta = make_synthetic_state_dict(arch, seed, num_classes, mean, std, **kw)
    blob, info = from_state_dict(meta["arch"], sd, bn_eps=meta["bn_eps"], mean=mean, std=std)
    info.update(seed=seed, source="synthetic")
    return blob, info


def pack_blob(arch: int, num_classes: int, specs, folded) -> bytes:
    n = len(specs)
    off = 32 + 48 * n
    off = (off + 63) // 64 * 64
    table, chunks = [], []
    for sp, (wq, bq) in zip(specs, folded):
        wb = (np.ascontiguousarray(wq, np.float32) if sp["kh"] == 0 else np.ascontiguousarray(wq, np.uint16)).tobytes()
        w_off = off
        off = (off + len(wb) + 63) // 64 * 64
        bb = np.ascontiguousarray(bq, np.float32).tobytes()
        b_off = off
        off = (off + len(bb) + 63) // 64 * 64
        table.append(struct.pack("<8I2Q", sp["cout"], sp["cin"], sp["kh"], sp["kw"], sp["stride"], sp["pad"], 0, 0,
                                 w_off, b_off))
        chunks.append((w_off, wb))
        chunks.append((b_off, bb))
    buf = bytearray(off)
    struct.pack_into("<8I", buf, 0, BLOB_MAGIC, 1, arch, num_classes, n, 0, 0, 0)
    for i, t in enumerate(table):
        buf[32 + 48 * i:32 + 48 * (i + 1)] = t
    for o, b in chunks:
        buf[o:o + len(b)] = b
    return bytes(buf)


# ---------------------------------------------------------------------------------------------------------
# Trained checkpoints: torch state_dict -> FAVW blob
# ---------------------------------------------------------------------------------------------------------
def _np(t):
    return np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t, dtype=np.float64)


def _fold_conv_bn(sd, conv, bn, eps):
    """OIHW conv weight (+ optional conv bias) followed by eval-mode BatchNorm -> (bf16 bits [O,kh,kw,I], fp32 bias[O]).
    The BN scale is multiplied into the weights BEFORE their bf16 rounding, the shift goes to the fp32 bias."""
    w = _np(sd[conv + ".weight"])
    gamma, beta = _np(sd[bn + ".weight"]), _np(sd[bn + ".bias"])
    mean, var = _np(sd[bn + ".running_mean"]), _np(sd[bn + ".running_var"])
    scale = gamma / np.sqrt(var + eps)
    cb = _np(sd[conv + ".bias"]) if conv + ".bias" in sd else 0.0
    wq = _bf16((w * scale[:, None, None, None]).transpose(0, 2, 3, 1).astype(np.float32))
    return wq, (beta + (cb - mean) * scale).astype(np.float32)


def _linear(sd, name, shape_ohwi=None):
    w = _np(sd[name + ".weight"])
    if w.ndim == 4:                                   # a convolution used as a linear map (ViT patch embedding)
        w = w.transpose(0, 2, 3, 1)
    else:
        w = w.reshape(w.shape[0], 1, 1, -1)
    b = _np(sd[name + ".bias"]) if name + ".bias" in sd else np.zeros(w.shape[0])
    return _bf16(w.astype(np.float32)), b.astype(np.float32)


def from_state_dict(arch, sd, num_classes: int | None = None, bn_eps: float = 1e-5, mean=DEFAULT_MEAN, std=DEFAULT_STD):
    """Converts a trained torch ``state_dict`` into the FAVW blob ``fav_load_weights`` takes -> (blob, info).

    * ``resnet50`` / ``resnet18_cifar``: torchvision naming (``conv1``, ``bn1``, ``layerL.B.convK`` / ``bnK``,
      ``layerL.B.downsample.0`` / ``.1``, ``fc``).  Every BatchNorm is folded in eval mode: scale into the bf16
      weights (before rounding), shift into the fp32 bias; the blob's layer order is stem, per block conv1, conv2,
      (conv3), (downsample), then fc.  ResNet-50 must be v1.5 (stride on the 3x3), as torchvision's is.
    * ``vit_b16`` / ``vit_tiny``: timm naming (``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``blocks.i.norm1``,
      ``attn.qkv``, ``attn.proj``, ``norm2``, ``mlp.fc1``, ``mlp.fc2``, ``norm``, ``head``); qkv rows are
      Q | K | V with head h in rows 64h..64h+63 of each, the class token is added into row 0 of the position table.
      The device MLP uses the erf form of GELU (torch.nn.GELU's default, what timm checkpoints are trained with)
      through a fixed polynomial for the normal CDF, within 8.6e-5 of it.
    """
    aid = ARCH_IDS[arch] if isinstance(arch, str) else int(arch)
    specs, folded = [], []
    if aid in VIT_CFG:
        cfg = VIT_CFG[aid]
        D, depth, P = cfg["dim"], cfg["depth"], cfg["patch"]
        pe = _np(sd["patch_embed.proj.weight"])
        if pe.shape != (D, 3, P, P):
            raise ValueError(f"patch_embed.proj.weight is {pe.shape}, expected {(D, 3, P, P)}")
        specs.append(dict(cout=D, cin=3, kh=P, kw=P, stride=P, pad=0)); folded.append(_linear(sd, "patch_embed.proj"))
        pos = _np(sd["pos_embed"]).reshape(-1, D).copy()
        pos[0] += _np(sd["cls_token"]).reshape(D)
        specs.append(dict(cout=pos.size, cin=0, kh=0, kw=0, stride=0, pad=0))
        folded.append((pos.reshape(-1).astype(np.float32), np.zeros(pos.size, np.float32)))

        def vec(name):
            specs.append(dict(cout=D, cin=0, kh=0, kw=0, stride=0, pad=0))
            folded.append((_np(sd[name + ".weight"]).astype(np.float32), _np(sd[name + ".bias"]).astype(np.float32)))

        def lin(name):
            wq, b = _linear(sd, name)
            specs.append(dict(cout=wq.shape[0], cin=wq.shape[3], kh=1, kw=1, stride=1, pad=0)); folded.append((wq, b))

        for i in range(depth):
            p = f"blocks.{i}."
            vec(p + "norm1"); lin(p + "attn.qkv"); lin(p + "attn.proj"); vec(p + "norm2"); lin(p + "mlp.fc1"); lin(p + "mlp.fc2")
        vec("norm"); lin("head")
        ncls = folded[-1][0].shape[0]
        ntok = pos.shape[0]
        hw = int(round((ntok - 1) ** 0.5)) * P
        info_extra = dict(in_hw=(hw, hw), n_tokens=ntok)
    else:
        ncls = int(_np(sd["fc.weight"]).shape[0])
        lspecs = layer_specs(aid, ncls)
        for sp, (conv, bn) in zip(lspecs, tv_names(aid, lspecs)):
            wq, b = _linear(sd, conv) if bn is None else _fold_conv_bn(sd, conv, bn, bn_eps)
            if wq.shape != (sp["cout"], sp["kh"], sp["kw"], sp["cin"]):
                raise ValueError(f"{sp['role']} of block {sp['block']}: checkpoint tensor is {wq.shape}, the architecture expects "
                                 f"{(sp['cout'], sp['kh'], sp['kw'], sp['cin'])} (ResNet-50 must be v1.5)")
            specs.append(sp); folded.append((wq, b))
        info_extra = {}
    if num_classes is not None and int(num_classes) != ncls:
        raise ValueError(f"checkpoint has {ncls} classes, {num_classes} requested")
    blob = pack_blob(aid, ncls, specs, folded)
    info = dict(arch=arch, num_classes=ncls, n_layers=len(specs), sha256=hashlib.sha256(blob).hexdigest(), mean=tuple(mean),
                std=tuple(std), source="state_dict", **info_extra)
    return blob, info
