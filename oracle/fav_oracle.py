"""CPU oracle for the failure-aware classification hot path (NumPy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing in ``failure_aware_vision_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED.  The reference (Indra-jith/failure-aware-vision) contains no
classifier, no ResNet, no dropout sampler and no confidence head (SURVEY.md §0,
§8a rows a4-a9 are all "build-defined"), so there is no reference file:line this
arithmetic can follow and no golden vector of the reference pins it.  What the
reference does fix is the seam the result plugs into:

* the scorer returns ``anomaly_score`` in [0, 1]
  (platform/backend/signal_analyzer.py:114-121) and a ``vision_status`` string
  (signal_analyzer.py:145-171);
* the consumer accepts ``(vision_status, anomaly_score | None, dt)``
  (platform/backend/trust_engine.py:139-149) and treats the score as a
  penalty-only sensor (trust_engine.py:192-200).

This file is therefore the specification itself, cross-checked against
``torch.nn.functional`` on CPU in ``tests/test_oracle.py`` so that it is not
self-referential, and its Philox generator is pinned by the Random123
known-answer vectors.

Numerical contract (mirrored by the HIP path):

* activations are bf16 at every layer boundary (round-to-nearest-even),
  weights are bf16 with the BatchNorm scale folded in, accumulation is fp32;
* epilogue order: ``((acc + bias) + residual)`` in fp32, ReLU, dropout
  (``x * scale`` or 0), one rounding to bf16;
* global average pool: sequential fp32 sum over (h, w) row-major, times
  fp32(1/HW), rounded to bf16;
* logits stay fp32; softmax, mean over samples, entropy are fp32;
* dropout masks: Philox4x32-10, key=(seed_lo, seed_hi),
  counter=(element_index//16, global_image_index, sample t, site); the four
  output words give sixteen 8-bit draws (little endian); keep iff draw >= thr,
  thr = round(p*256), survivors scaled by fp32(1/(1 - thr/256)).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# ----------------------------------------------------------------------------
# bf16 helpers
# ----------------------------------------------------------------------------

def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 to the nearest bf16 (ties to even); result kept as fp32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32).reshape(x.shape)


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 (already bf16-representable) -> uint16 bit pattern."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


# ----------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., SC'11; Random123).  Counter-based, so the GPU
# epilogue and this file produce the same bits for the same (key, counter).
# ----------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised Philox4x32-10.  c* are uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, np.uint32), np.asarray(c1, np.uint32),
        np.asarray(c2, np.uint32), np.asarray(c3, np.uint32))
    c0 = c0.astype(np.uint64); c1 = c1.astype(np.uint64)
    c2 = c2.astype(np.uint64); c3 = c3.astype(np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for rnd in range(10):
        ka = np.uint64((k0 + rnd * _PHILOX_W0) & 0xFFFFFFFF)
        kb = np.uint64((k1 + rnd * _PHILOX_W1) & 0xFFFFFFFF)
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ ka, lo1, hi0 ^ c3 ^ kb, lo0
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def dropout_threshold(p: float) -> int:
    """8-bit drop threshold: an element is dropped iff its draw < thr."""
    return int(round(float(p) * 256.0))


def dropout_scale(thr: int) -> np.float32:
    return np.float32(1.0 / (1.0 - thr / 256.0)) if thr > 0 else np.float32(1.0)


def dropout_keep(seed: int, t: int, site: int, img_ids: np.ndarray, n_elem: int, thr: int) -> np.ndarray:
    """Boolean keep mask [len(img_ids), n_elem] (n_elem multiple of 16).  One Philox
    call per 16 elements: byte j (little endian over the four words) is element j's draw."""
    assert n_elem % 16 == 0
    chunks = np.arange(n_elem // 16, dtype=np.uint32)[None, :]
    imgs = np.asarray(img_ids, dtype=np.uint32)[:, None]
    w = philox4x32_10(chunks, imgs, np.uint32(t), np.uint32(site),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    draws = np.empty((imgs.shape[0], n_elem // 16, 16), dtype=np.uint32)
    for q in range(4):
        for b in range(4):
            draws[:, :, 4 * q + b] = (w[q] >> np.uint32(8 * b)) & np.uint32(0xFF)
    return (draws >= np.uint32(thr)).reshape(imgs.shape[0], n_elem)


# ----------------------------------------------------------------------------
# Weight blob ("FAVW", version 1) — the same bytes fav_load_weights() takes.
# ----------------------------------------------------------------------------
BLOB_MAGIC = 0x57564146  # "FAVW" little endian
ARCH_RESNET18_CIFAR = 0
ARCH_RESNET50 = 1


@dataclass
class ConvLayer:
    cout: int
    cin: int
    kh: int
    kw: int
    stride: int
    pad: int
    w: np.ndarray  # fp32 holding bf16 values, [cout, kh, kw, cin]
    b: np.ndarray  # fp32 [cout]


@dataclass
class Model:
    arch: int
    num_classes: int
    layers: list = field(default_factory=list)


def parse_blob(blob: bytes) -> Model:
    magic, version, arch, ncls, nlayers = struct.unpack_from("<5I", blob, 0)
    if magic != BLOB_MAGIC or version != 1:
        raise ValueError("not a FAVW v1 blob")
    m = Model(arch, ncls)
    for i in range(nlayers):
        cout, cin, kh, kw, stride, pad, _, _, w_off, b_off = struct.unpack_from("<8I2Q", blob, 32 + 48 * i)
        if kh == 0:   # a pair of fp32 vectors (LayerNorm gamma/beta, ViT position table)
            m.layers.append(ConvLayer(cout, 0, 0, 0, 0, 0, np.frombuffer(blob, np.float32, cout, w_off).copy(),
                                      np.frombuffer(blob, np.float32, cout, b_off).copy()))
            continue
        n = cout * kh * kw * cin
        w = bf16_from_bits(np.frombuffer(blob, np.uint16, n, w_off)).reshape(cout, kh, kw, cin)
        b = np.frombuffer(blob, np.float32, cout, b_off).copy()
        m.layers.append(ConvLayer(cout, cin, kh, kw, stride, pad, w.copy(), b))
    return m


# ----------------------------------------------------------------------------
# Layer primitives (NHWC)
# ----------------------------------------------------------------------------

def sanitize_pixels(x: np.ndarray) -> np.ndarray:
    """fp32 frames as the library takes them in (fav_kernels.hpp: fav_sanitize_px): NaN -> 0, then clamped to [-64, 64];
    pixels in [0, 1] pass unchanged.  The reference answers a garbage frame with a status, never an exception
    (signal_analyzer.py:145-171); here it cannot poison the network with non-finite values."""
    x = np.asarray(x, np.float32)
    return np.clip(np.where(np.isnan(x), np.float32(0), x), np.float32(-64), np.float32(64)).astype(np.float32)


def normalize_input(x, mean, inv_std) -> np.ndarray:
    """[B,H,W,3] uint8 (0..255) or fp32 in [0,1] -> normalised bf16 values."""
    if x.dtype == np.uint8:
        x = x.astype(np.float32) * np.float32(1.0 / 255.0)
    else:
        x = sanitize_pixels(x)
    x = x.astype(np.float32)
    y = (x - np.asarray(mean, np.float32)) * np.asarray(inv_std, np.float32)
    return bf16_round(y)


def _im2col(x: np.ndarray, kh: int, kw: int, stride: int, pad: int):
    b, h, w, c = x.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (w + 2 * pad - kw) // stride + 1
    xp = np.zeros((b, h + 2 * pad, w + 2 * pad, c), np.float32)
    xp[:, pad:pad + h, pad:pad + w] = x
    sb, sh, sw, sc = xp.strides
    patches = np.lib.stride_tricks.as_strided(
        xp, (b, ho, wo, kh, kw, c), (sb, sh * stride, sw * stride, sh, sw, sc), writeable=False)
    return patches.reshape(b * ho * wo, kh * kw * c), ho, wo


_EXACT_LIB = None


def _exact_lib(optional=False):
    """oracle/_build/libfav_exact.so: the accumulator in the HIP kernel's exact-mode order.  optional=True: None
    instead of a build attempt when the library is absent (callers then use their NumPy form)."""
    global _EXACT_LIB
    if _EXACT_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libfav_exact.so")
        if not os.path.exists(path):
            if optional:
                return None
            import subprocess
            subprocess.check_call(["make", "-C", os.path.dirname(os.path.abspath(__file__))])
        lib = ctypes.CDLL(path)
        lib.fav_epilogue_bf16.restype = None
        lib.fav_epilogue_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_long,
                                          ctypes.c_int]
        lib.fav_bf16mfma_conv_acc_order.restype = ctypes.c_int
        lib.fav_bf16mfma_conv_acc_order.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 9 + [ctypes.c_void_p]
        for fn in (lib.fav_exact_conv_acc, lib.fav_bf16mfma_conv_acc):
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 9
        for fn in (lib.fav_expf_arr, lib.fav_gelu_arr):
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        lib.fav_layernorm_rows.restype = None
        lib.fav_layernorm_rows.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_float]
        lib.fav_attn_softmax_rows.restype = None
        lib.fav_attn_softmax_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
        lib.fav_bf16mfma_replay.restype = None
        lib.fav_bf16mfma_replay.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int]
        _EXACT_LIB = lib
    return _EXACT_LIB


def conv_acc_exact(x: np.ndarray, w: np.ndarray, kh: int, kw: int, stride: int, pad: int, mode=True) -> np.ndarray:
    """Order-exact fp32 accumulator (oracle/fav_exact.c).  x [B,H,W,C], w [N,kh,kw,C], C % 64 == 0.
    mode True  : k-ordered fmaf chain of the device's FAV_MATH_F32_EXACT mode;
    mode "mfma": the bit-exact model of v_mfma_f32_16x16x32_bf16, i.e. the production mode."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    b, h, ww, c = x.shape
    n = w.shape[0]
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (ww + 2 * pad - kw) // stride + 1
    out = np.empty((b, ho, wo, n), np.float32)
    fn = _exact_lib().fav_bf16mfma_conv_acc if mode == "mfma" else _exact_lib().fav_exact_conv_acc
    rc = fn(x.ctypes.data, w.ctypes.data, out.ctypes.data, b, h, ww, c, n, kh, kw, stride, pad)
    if rc != 0:
        raise RuntimeError(f"fav_exact_conv_acc failed ({rc})")
    return out


def conv_acc(x: np.ndarray, L: ConvLayer, exact: bool = False) -> np.ndarray:
    """fp32 accumulator of the convolution, [B,Ho,Wo,Cout] (no bias).
    exact=False: im2col + BLAS sgemm (any summation order).
    exact=True : the HIP kernel's exact-mode order, bit-reproducible (fav_exact.c);
                 a Cin=3 stem goes through the zero-padded im2col matrix exactly as
                 the device does (K padded to a multiple of 64)."""
    if not exact:
        cols, ho, wo = _im2col(x, L.kh, L.kw, L.stride, L.pad)
        acc = cols @ L.w.reshape(L.cout, -1).T
        return acc.reshape(x.shape[0], ho, wo, L.cout).astype(np.float32)
    if L.cin % 64 == 0:
        return conv_acc_exact(x, L.w, L.kh, L.kw, L.stride, L.pad, exact)
    cols, ho, wo = _im2col(x, L.kh, L.kw, L.stride, L.pad)
    k = cols.shape[1]
    kpad = (k + 63) // 64 * 64
    a1 = np.zeros((x.shape[0], ho, wo, kpad), np.float32)
    a1[..., :k] = cols.reshape(x.shape[0], ho, wo, k)
    wp = np.zeros((L.cout, 1, 1, kpad), np.float32)
    wp[:, 0, 0, :k] = L.w.reshape(L.cout, k)
    return conv_acc_exact(a1, wp, 1, 1, 1, 0, exact)


def epilogue(acc, bias, res=None, relu=True, keep=None, scale=np.float32(1.0)) -> np.ndarray:
    """((acc + bias) + residual), ReLU, dropout, one rounding to bf16.  `epilogue_numpy` is the specification; the C
    restatement (fav_exact.c: fav_epilogue_bf16, the same fp32 operations in the same order, multithreaded) is used when
    the library is built - tests/test_oracle.py compares the two."""
    lib = _exact_lib(optional=True)
    if lib is None:
        return epilogue_numpy(acc, bias, res, relu, keep, scale)
    acc = np.ascontiguousarray(acc, np.float32)
    b32 = np.ascontiguousarray(bias, np.float32)
    c = acc.shape[-1]
    assert b32.shape == (c,)
    r32 = None if res is None else np.ascontiguousarray(np.broadcast_to(res, acc.shape), np.float32)
    k8 = None if keep is None else np.ascontiguousarray(np.asarray(keep).reshape(acc.shape), np.uint8)
    out = np.empty_like(acc)
    lib.fav_epilogue_bf16(acc.ctypes.data, b32.ctypes.data, None if r32 is None else r32.ctypes.data,
                          None if k8 is None else k8.ctypes.data, float(scale), 1 if relu else 0, out.ctypes.data,
                          acc.size // c, c)
    return out


def epilogue_numpy(acc, bias, res=None, relu=True, keep=None, scale=np.float32(1.0)) -> np.ndarray:
    y = acc + bias.astype(np.float32)
    if res is not None:
        y = y + res
    if relu:
        y = np.maximum(y, np.float32(0.0))
    if keep is not None:
        y = np.where(keep.reshape(y.shape), y * scale, np.float32(0.0)).astype(np.float32)
    return bf16_round(y)


def maxpool3x3s2(x: np.ndarray) -> np.ndarray:
    b, h, w, c = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    xp = np.full((b, h + 2, w + 2, c), -np.inf, np.float32)
    xp[:, 1:1 + h, 1:1 + w] = x
    out = np.full((b, ho, wo, c), -np.inf, np.float32)
    for r in range(3):
        for s in range(3):
            out = np.maximum(out, xp[:, r:r + 2 * ho:2, s:s + 2 * wo:2])
    return out


def global_avgpool(x: np.ndarray) -> np.ndarray:
    """Sequential fp32 sum over (h, w) row-major, * fp32(1/HW); NOT yet rounded."""
    b, h, w, c = x.shape
    acc = np.zeros((b, c), np.float32)
    for i in range(h):
        for j in range(w):
            acc = acc + x[:, i, j, :]
    return acc * np.float32(1.0 / (h * w))


# ----------------------------------------------------------------------------
# Networks.  A network is: stem, blocks[0..n), pool, fc.  Dropout site s < n is
# the output of block s; site n is the pooled feature vector (fc input).
# ----------------------------------------------------------------------------
_RESNET50_CFG = dict(block="bottleneck", depths=(3, 4, 6, 3), planes=(64, 128, 256, 512), stem="imagenet")
_RESNET18_CFG = dict(block="basic", depths=(2, 2, 2, 2), planes=(64, 128, 256, 512), stem="cifar")


def arch_cfg(arch: int) -> dict:
    return _RESNET50_CFG if arch == ARCH_RESNET50 else _RESNET18_CFG


def block_table(arch: int):
    """[(n_convs_in_main_path, has_downsample)] per block, in blob order."""
    cfg = arch_cfg(arch)
    nmain = 3 if cfg["block"] == "bottleneck" else 2
    exp = 4 if cfg["block"] == "bottleneck" else 1
    out, inplanes = [], 64
    for li, (d, p) in enumerate(zip(cfg["depths"], cfg["planes"])):
        for bi in range(d):
            stride = 2 if (bi == 0 and li > 0) else 1
            ds = (bi == 0) and (stride != 1 or inplanes != p * exp)
            out.append((nmain, ds))
            inplanes = p * exp
    return out


def n_sites(arch: int) -> int:
    return len(block_table(arch)) + 1


class OracleNet:
    def __init__(self, model: Model, exact: bool = False):
        self.m = model
        self.exact = exact
        self.trace = None   # set to a list to record (layer, input, residual, relu, output) per convolution
        self.cfg = arch_cfg(model.arch)
        self.blocks = []
        idx = 1
        for nmain, ds in block_table(model.arch):
            main = model.layers[idx:idx + nmain]; idx += nmain
            down = None
            if ds:
                down = model.layers[idx]; idx += 1
            self.blocks.append((main, down))
        self.fc = model.layers[idx]
        assert idx + 1 == len(model.layers)
        self.nb = len(self.blocks)

    def _rec(self, L, x, res, relu, y):
        if self.trace is not None:
            self.trace.append((L, x, res, relu, y))
        return y

    # -- stages -------------------------------------------------------------
    def stem(self, xn):
        y = epilogue(conv_acc(xn, self.m.layers[0], self.exact), self.m.layers[0].b)
        self._rec(self.m.layers[0], xn, None, True, y)
        if self.cfg["stem"] == "imagenet":
            y = maxpool3x3s2(y)
        return y

    def block(self, i, x, keep=None, scale=np.float32(1.0)):
        main, down = self.blocks[i]
        h = x
        for L in main[:-1]:
            h = self._rec(L, h, None, True, epilogue(conv_acc(h, L, self.exact), L.b))
        idn = x if down is None else self._rec(down, x, None, False, epilogue(conv_acc(x, down, self.exact), down.b, relu=False))
        L = main[-1]
        y = epilogue(conv_acc(h, L, self.exact), L.b, res=idn, keep=keep, scale=scale)
        return self._rec(L, h, idn, True, y) if keep is None else y

    def pool(self, x, keep=None, scale=np.float32(1.0)):
        y = global_avgpool(x)
        if keep is not None:
            y = np.where(keep, y * scale, np.float32(0.0)).astype(np.float32)
        return bf16_round(y)

    def logits(self, feat):
        if self.exact:
            acc = conv_acc_exact(feat[:, None, None, :], self.fc.w, 1, 1, 1, 0, self.exact)[:, 0, 0, :]
            return acc + self.fc.b
        return (feat @ self.fc.w.reshape(self.fc.cout, -1).T).astype(np.float32) + self.fc.b

    # -- full forward -------------------------------------------------------
    def forward_logits(self, xn, img_ids=None, n_samples=1, site_mask=0, p=0.0, seed=0):
        """Returns logits [T, B, C].  site_mask==0 or p==0 -> deterministic, T forced to 1."""
        b = xn.shape[0]
        img_ids = np.arange(b) if img_ids is None else np.asarray(img_ids)
        thr = dropout_threshold(p)
        if site_mask == 0 or thr == 0:
            site_mask, n_samples, thr = 0, 1, 0
        scale = dropout_scale(thr)
        first = min((s for s in range(self.nb + 1) if site_mask >> s & 1), default=self.nb + 1)

        def run_from(stage, act, t):
            # stage s in [0, nb): input of block s is `act`; stage nb: act is pre-pool map
            for i in range(stage, self.nb):
                act = self._block_with_site(i, act, t, img_ids, site_mask, seed, thr, scale)
            keep = None
            if site_mask >> self.nb & 1:
                keep = dropout_keep(seed, t, self.nb, img_ids, act.shape[-1], thr)
            return self.logits(self.pool(act, keep, scale))

        # deterministic prefix, computed once: stem + blocks up to and including block
        # `first` (or through the pool when the first site is the pooled vector).  The
        # first site's dropout is applied per sample to that cached, already bf16-rounded
        # tensor ("entry dropout": x*scale re-rounded to bf16); later sites are fused
        # into their producer's epilogue before its single rounding.
        act = self.stem(xn)
        if first > self.nb:  # no dropout at all
            return run_from(0, act, 0)[None]
        npre = min(first + 1, self.nb)  # blocks computed once
        for i in range(npre):
            act = self.block(i, act)
        if first == self.nb:
            act = self.pool(act)        # [B, C], rounded

        def entry(a, t):
            keep = dropout_keep(seed, t, first, img_ids, int(np.prod(a.shape[1:])), thr)
            return bf16_round(np.where(keep.reshape(a.shape), a * scale, np.float32(0.0)).astype(np.float32))

        out = []
        for t in range(n_samples):
            if first == self.nb:
                out.append(self.logits(entry(act, t)))
            else:
                out.append(run_from(npre, entry(act, t), t))
        return np.stack(out)

    def _block_with_site(self, i, act, t, img_ids, site_mask, seed, thr, scale):
        if not (site_mask >> i & 1):
            return self.block(i, act)
        # need the output element count: same spatial dims as the last conv's output
        main, _ = self.blocks[i]
        ho = act.shape[1]; wo = act.shape[2]
        for L in main:
            ho = (ho + 2 * L.pad - L.kh) // L.stride + 1
            wo = (wo + 2 * L.pad - L.kw) // L.stride + 1
        keep = dropout_keep(seed, t, i, img_ids, ho * wo * main[-1].cout, thr)
        return self.block(i, act, keep, scale)


# ----------------------------------------------------------------------------
# Confidence head, failure detector, score adapter
# ----------------------------------------------------------------------------
CONF_MAX_SOFTMAX = 0
CONF_ENTROPY = 1



# ----------------------------------------------------------------------------
# ViT-B/16 (BASELINE configs[4]).  Build-defined like the rest (PARITY UNPINNED): pre-norm
# encoder, class token, learned positions, GELU MLP, bf16 at every tensor boundary,
# fp32 accumulation / statistics.  exp, GELU, LayerNorm and the attention softmax are the
# exact fp32 operation sequences of oracle/fav_exact.c, which the HIP kernels repeat.
# ----------------------------------------------------------------------------
ARCH_VIT_B16 = 2
ARCH_VIT_TINY = 3
VIT_CFG = {ARCH_VIT_B16: dict(dim=768, depth=12, heads=12, mlp=3072, patch=16),
           ARCH_VIT_TINY: dict(dim=128, depth=2, heads=2, mlp=256, patch=16)}
LN_EPS = np.float32(1e-6)


def _c_unary(name, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    getattr(_exact_lib(), name)(x.ctypes.data, y.ctypes.data, x.size)
    return y


def expf_exact(x):
    return _c_unary("fav_expf_arr", x)


def gelu_exact(x):
    return _c_unary("fav_gelu_arr", x)


def layernorm_exact(x, gamma, beta, eps=LN_EPS):
    """fp32 LayerNorm over the last axis in the device's reduction order (not yet rounded to bf16)."""
    x = np.ascontiguousarray(x, np.float32)
    d = x.shape[-1]
    y = np.empty_like(x)
    g, b = np.ascontiguousarray(gamma, np.float32), np.ascontiguousarray(beta, np.float32)
    _exact_lib().fav_layernorm_rows(x.ctypes.data, g.ctypes.data, b.ctypes.data, y.ctypes.data, x.size // d, d, float(eps))
    return y


def attn_softmax_exact(s):
    """softmax(s / 8) over the last axis of RAW attention scores, in the device's operation order (fav_attn_softmax_rows)."""
    s = np.ascontiguousarray(s, np.float32)
    p = np.empty_like(s)
    _exact_lib().fav_attn_softmax_rows(s.ctypes.data, p.ctypes.data, s.size // s.shape[-1], s.shape[-1])
    return p


def gemm_acc(a, w, exact=False):
    """fp32 accumulator a[M,K] @ w[N,K]^T in the chosen summation model (K zero-padded to 64)."""
    a = np.ascontiguousarray(a, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    if not exact:
        return (a @ w.T).astype(np.float32)
    m, k = a.shape
    kp = (k + 63) // 64 * 64
    if kp != k:
        a = np.concatenate([a, np.zeros((m, kp - k), np.float32)], axis=1)
        w = np.concatenate([w, np.zeros((w.shape[0], kp - k), np.float32)], axis=1)
    return conv_acc_exact(a.reshape(1, m, 1, kp), w.reshape(w.shape[0], 1, 1, kp), 1, 1, 1, 0, exact).reshape(m, w.shape[0])


def attn_key_order(t):
    """Key held by each k slot of the second product, O = P V: the device feeds the score accumulators (4 consecutive keys per
    lane group g and 16-key tile) straight back as the MFMA operand, so slot 8g + j of 32-key block b holds key
    32b + 16 (j >> 2) + 4g + (j & 3).  Returns the keys in slot order over ceil(t / 32) blocks (keys >= t are zero padding)."""
    slot = np.arange((t + 31) // 32 * 32)
    b, g, j = slot >> 5, (slot >> 3) & 3, slot & 7
    return 32 * b + 16 * (j >> 2) + 4 * g + (j & 3)


def attention(qkv, heads, exact=False):
    """qkv [B, T, 3D] bf16 values -> [B, T, D] bf16: per head softmax(Q K^T / 8) V with fp32
    scores, probabilities rounded to bf16 before the second product, whose keys go through the
    summation model in the device's slot order (attn_key_order)."""
    b, t, d3 = qkv.shape
    d = d3 // 3
    out = np.empty((b, t, d), np.float32)
    order = attn_key_order(t)
    pp = np.zeros((t, order.size), np.float32)
    vp = np.zeros((order.size, 64), np.float32)
    for i in range(b):
        for h in range(heads):
            q = qkv[i, :, h * 64:(h + 1) * 64]
            k = qkv[i, :, d + h * 64:d + (h + 1) * 64]
            v = qkv[i, :, 2 * d + h * 64:2 * d + (h + 1) * 64]
            pp[:, :t] = bf16_round(attn_softmax_exact(gemm_acc(q, k, exact)))     # raw scores: the 1 / 8 is inside the exponential's constant
            vp[:t] = v
            out[i, :, h * 64:(h + 1) * 64] = bf16_round(gemm_acc(pp[:, order], np.ascontiguousarray(vp[order].T), exact))
    return out


class VitNet:
    def __init__(self, model: Model, exact=False):
        self.m, self.exact = model, exact
        self.cfg = VIT_CFG[model.arch]
        self.trace = None

    def _lin(self, x, L, res=None, act=None):
        y = gemm_acc(x.reshape(-1, x.shape[-1]), L.w.reshape(L.cout, -1), self.exact) + L.b
        if res is not None:
            y = y + res.reshape(-1, L.cout)
        if act == "gelu":
            y = gelu_exact(y)
        return y.reshape(x.shape[:-1] + (L.cout,)).astype(np.float32)

    def _ln(self, x, L):
        return bf16_round(layernorm_exact(x, L.w, L.b))

    def forward_logits(self, xn, *unused, **unused_kw):
        c, Ls = self.cfg, self.m.layers
        b, hh, ww, _ = xn.shape
        p, d = c["patch"], c["dim"]
        gh, gw = hh // p, ww // p
        patches = xn.reshape(b, gh, p, gw, p, 3).transpose(0, 1, 3, 2, 4, 5).reshape(b, gh * gw, p * p * 3)
        emb = bf16_round(self._lin(patches, Ls[0]))
        pos = Ls[1].w.reshape(-1, d)
        if pos.shape[0] != gh * gw + 1:
            raise ValueError("position table does not match the input size")
        x = np.empty((b, gh * gw + 1, d), np.float32)
        x[:, 0] = pos[0]
        x[:, 1:] = emb + pos[1:]
        x = bf16_round(x)
        li = 2
        for _ in range(c["depth"]):
            ln1, qkv, proj, ln2, fc1, fc2 = Ls[li:li + 6]
            li += 6
            y = self._ln(x, ln1)
            a = attention(bf16_round(self._lin(y, qkv)), c["heads"], self.exact)
            x = bf16_round(self._lin(a, proj, res=x))
            y = self._ln(x, ln2)
            hdn = bf16_round(self._lin(y, fc1, act="gelu"))
            x = bf16_round(self._lin(hdn, fc2, res=x))
            if self.trace is not None:
                self.trace.append(x.copy())
        y = self._ln(x[:, 0], Ls[li])
        return self._lin(y, Ls[li + 1])[None]


def mean_softmax(logits: np.ndarray, temperature: float = 1.0) -> np.ndarray:
    """logits [T,B,C] fp32 -> mean over T of softmax(z * fp32(1/temp)), fp32 [B,C]."""
    z = logits.astype(np.float32) * np.float32(1.0 / temperature)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z).astype(np.float32)
    p = e / e.sum(axis=-1, keepdims=True, dtype=np.float32)
    acc = np.zeros(p.shape[1:], np.float32)
    for t in range(p.shape[0]):
        acc = acc + p[t]
    return acc * np.float32(1.0 / p.shape[0])


def confidence_head(logits: np.ndarray, temperature: float = 1.0, kind: int = CONF_MAX_SOFTMAX):
    """-> (labels int32[B], confidence fp32[B], pbar fp32[B,C]).  argmax tie -> lowest index."""
    pbar = mean_softmax(logits, temperature)
    labels = pbar.argmax(axis=-1).astype(np.int32)
    if kind == CONF_MAX_SOFTMAX:
        conf = pbar.max(axis=-1)
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            plogp = np.where(pbar > 0, pbar * np.log(pbar), np.float32(0.0)).astype(np.float32)
        h = -plogp.sum(axis=-1, dtype=np.float32)
        conf = np.float32(1.0) - h * np.float32(1.0 / np.log(pbar.shape[-1]))
    return labels, conf.astype(np.float32), pbar


def failure_detect(conf: np.ndarray, tau: float):
    """fail = conf < tau ; anomaly_score = 1 - conf, clamped to [0,1]
    (same range contract as signal_analyzer.py:121)."""
    fail = (conf < np.float32(tau)).astype(np.uint8)
    score = np.clip(np.float32(1.0) - conf, np.float32(0.0), np.float32(1.0)).astype(np.float32)
    return fail, score


# ----------------------------------------------------------------------------
# End-to-end classify (what fav_classify computes)
# ----------------------------------------------------------------------------
@dataclass
class ClassifyConfig:
    mean: tuple = (0.485, 0.456, 0.406)
    std: tuple = (0.229, 0.224, 0.225)
    n_samples: int = 1
    site_mask: int = 0
    p: float = 0.0
    seed: int = 0
    temperature: float = 1.0
    conf_kind: int = CONF_MAX_SOFTMAX
    tau: float = 0.5
    exact: object = False   # True: reproduce the device's FAV_MATH_F32_EXACT mode bit for bit;
                            # "mfma": reproduce the production bf16-MFMA mode bit for bit


def inv_std32(std):
    return tuple(np.float32(1.0) / np.float32(s) for s in std)


def classify(model: Model, images: np.ndarray, cfg: ClassifyConfig, img_ids=None, return_logits=False):
    net = VitNet(model, exact=cfg.exact) if model.arch in VIT_CFG else OracleNet(model, exact=cfg.exact)
    xn = normalize_input(images, cfg.mean, inv_std32(cfg.std))
    lg = net.forward_logits(xn, img_ids, cfg.n_samples, cfg.site_mask, cfg.p, cfg.seed)
    labels, conf, pbar = confidence_head(lg, cfg.temperature, cfg.conf_kind)
    if return_logits:
        return labels, conf, lg, pbar
    return labels, conf


def gaussian_noise_corrupt(images01: np.ndarray, severity: int, seed: int) -> np.ndarray:
    """ImageNet-C style gaussian_noise on [0,1] fp32 pixels (external convention:
    sigma = 0.08/0.12/0.18/0.26/0.38 for severity 1..5), clipped to [0,1]."""
    sigma = (0.08, 0.12, 0.18, 0.26, 0.38)[severity - 1]
    rng = np.random.default_rng(seed)
    n = rng.standard_normal(images01.shape, dtype=np.float32) * np.float32(sigma)
    return np.clip(images01.astype(np.float32) + n, 0.0, 1.0).astype(np.float32)
