"""A plain fp32 PyTorch model of the classifier: what a PyTorch-CPU user would write.

TEST INFRASTRUCTURE ONLY - PARITY UNPINNED (see fav_oracle.py header): the reference lists torch at
requirements.txt:1-2 and never imports it, so there is no reference model to follow; this is the textbook
torchvision-style ResNet (v1.5: stride on the 3x3) with BatchNorm layers un-folded, fp32 weights and activations and
no bf16 anywhere.  ``weights.make_synthetic_state_dict`` produces exactly its ``state_dict``.

Two jobs: (1) the "stated tolerance against a pure fp32 PyTorch model" (tests/golden/make_fp32_module_fixture.py and
the GPU test that compares the production mode with it); (2) the ``cpu_baseline`` of bench.py: the headline
algorithm (deterministic prefix once, T dropout samples of the suffix, mean of softmax) as a PyTorch user would run
it - ``F.dropout`` masks from torch's own generator, no rounding work the GPU path's contract asks for.

Never imported by ``failure_aware_vision_amd``.
"""
import numpy as np
import torch
import torch.nn as nn


class _Bottleneck(nn.Module):
    def __init__(self, inpl, pl, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, pl, 1, bias=False); self.bn1 = nn.BatchNorm2d(pl)
        self.conv2 = nn.Conv2d(pl, pl, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(pl)   # v1.5: stride on the 3x3
        self.conv3 = nn.Conv2d(pl, pl * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(pl * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inpl, pl * 4, 1, stride, bias=False), nn.BatchNorm2d(pl * 4)) if down else None

    def forward(self, x):                                  # in-place add and ReLU, as torchvision's Bottleneck (ReLU(inplace=True), out += identity)
        idn = x if self.downsample is None else self.downsample(x)
        y = torch.relu_(self.bn1(self.conv1(x)))
        y = torch.relu_(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += idn
        return torch.relu_(y)


class _Basic(nn.Module):
    def __init__(self, inpl, pl, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, pl, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(pl)
        self.conv2 = nn.Conv2d(pl, pl, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(pl)
        self.downsample = nn.Sequential(nn.Conv2d(inpl, pl, 1, stride, bias=False), nn.BatchNorm2d(pl)) if down else None

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = torch.relu_(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += idn
        return torch.relu_(y)


class ResNet(nn.Module):
    def __init__(self, arch="resnet50", num_classes=1000):
        super().__init__()
        bott = arch == "resnet50"
        depths = (3, 4, 6, 3) if bott else (2, 2, 2, 2)
        exp = 4 if bott else 1
        self.imagenet = bott
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False) if bott else nn.Conv2d(3, 64, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inpl = 64
        for li, (d, pl) in enumerate(zip(depths, (64, 128, 256, 512))):
            blocks = []
            for bi in range(d):
                s = 2 if (bi == 0 and li > 0) else 1
                down = bi == 0 and (s != 1 or inpl != pl * exp)
                blocks.append((_Bottleneck if bott else _Basic)(inpl, pl, s, down))
                inpl = pl * exp
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.fc = nn.Linear(inpl, num_classes)

    def forward(self, x):
        x = torch.relu_(self.bn1(self.conv1(x)))
        if self.imagenet:
            x = nn.functional.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return self.fc(x.mean(dim=(2, 3)))



def load_synthetic(arch="resnet50", seed=1, num_classes=None):
    """The synthetic checkpoint as an nn.Module in eval mode -> (net, meta)."""
    from failure_aware_vision_amd import weights
    sd, meta = weights.make_synthetic_state_dict(arch, seed=seed, num_classes=num_classes)
    net = ResNet(arch if isinstance(arch, str) else {0: "resnet18_cifar", 1: "resnet50"}[arch], meta["num_classes"]).eval()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps = meta["bn_eps"]
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return net, meta


def blocks_of(net):
    return [b for i in range(1, 5) for b in getattr(net, f"layer{i}")]


def fuse_for_inference(net):
    """What a PyTorch user does before serving a ResNet: every conv + eval-mode BatchNorm pair becomes one fp32
    convolution with a bias (torch.nn.utils.fusion.fuse_conv_bn_eval).  Still fp32 everywhere, no bf16."""
    import copy
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    net = copy.deepcopy(net).eval()
    net.conv1 = fuse_conv_bn_eval(net.conv1, net.bn1); net.bn1 = nn.Identity()
    for b in blocks_of(net):
        for k in (1, 2, 3):
            if hasattr(b, f"conv{k}"):
                setattr(b, f"conv{k}", fuse_conv_bn_eval(getattr(b, f"conv{k}"), getattr(b, f"bn{k}")))
                setattr(b, f"bn{k}", nn.Identity())
        if b.downsample is not None:
            b.downsample = nn.Sequential(fuse_conv_bn_eval(b.downsample[0], b.downsample[1]), nn.Identity())
    return net.to(memory_format=torch.channels_last)


@torch.no_grad()
def mc_dropout_probs(net, x_nchw, n_samples, site_mask, p, chunk=96):
    """Mean over T dropout samples of softmax(logits) -> [B, classes].  Sites as in the GPU path: bit s = output of
    residual block s, bit n_blocks = pooled features; everything up to and including the first site's block runs
    once, the T samples of the rest run stacked, `chunk` virtual frames at a time (one stacked batch of T x B frames
    does not fit the caches of a CPU; F.dropout in place draws torch's own masks)."""
    blocks = blocks_of(net)
    nb = len(blocks)
    x = torch.relu_(net.bn1(net.conv1(x_nchw)))
    if net.imagenet:
        x = nn.functional.max_pool2d(x, 3, 2, 1)
    sites = [s for s in range(nb + 1) if site_mask >> s & 1]
    if not sites or p <= 0 or n_samples <= 1:
        for b in blocks:
            x = b(x)
        return torch.softmax(net.fc(x.mean(dim=(2, 3))), dim=1)
    first = sites[0]
    for i in range(min(first + 1, nb)):
        x = blocks[i](x)
    bsz = x.shape[0]
    drop = lambda t: nn.functional.dropout(t, p, training=True, inplace=True)
    probs = torch.zeros(bsz, net.fc.out_features)
    per = max(1, chunk // n_samples) if chunk < n_samples * bsz else bsz     # frames whose T samples run together
    for f0 in range(0, bsz, per):
        xs = x[f0:f0 + per].repeat(n_samples, 1, 1, 1)           # T copies of the cached prefix output
        if first < nb:
            xs = drop(xs)
        for i in range(first + 1, nb):
            xs = blocks[i](xs)
            if site_mask >> i & 1:
                xs = drop(xs)
        f = xs.mean(dim=(2, 3))
        if site_mask >> nb & 1:
            f = drop(f)
        nf = xs.shape[0] // n_samples
        probs[f0:f0 + nf] = torch.softmax(net.fc(f), dim=1).reshape(n_samples, nf, -1).mean(dim=0)
    return probs
