"""Minimal torch definitions with the naming of the usual trained checkpoints (torchvision ResNet, timm
VisionTransformer) - test infrastructure for weights.from_state_dict; torchvision and timm are not installed here."""
import torch
import torch.nn as nn


class _Bottleneck(nn.Module):
    def __init__(self, inpl, pl, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, pl, 1, bias=False); self.bn1 = nn.BatchNorm2d(pl)
        self.conv2 = nn.Conv2d(pl, pl, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(pl)   # v1.5: stride on the 3x3
        self.conv3 = nn.Conv2d(pl, pl * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(pl * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inpl, pl * 4, 1, stride, bias=False), nn.BatchNorm2d(pl * 4)) if down else None

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        return torch.relu(self.bn3(self.conv3(y)) + idn)


class _Basic(nn.Module):
    def __init__(self, inpl, pl, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, pl, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(pl)
        self.conv2 = nn.Conv2d(pl, pl, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(pl)
        self.downsample = nn.Sequential(nn.Conv2d(inpl, pl, 1, stride, bias=False), nn.BatchNorm2d(pl)) if down else None

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = torch.relu(self.bn1(self.conv1(x)))
        return torch.relu(self.bn2(self.conv2(y)) + idn)


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
        x = torch.relu(self.bn1(self.conv1(x)))
        if self.imagenet:
            x = nn.functional.max_pool2d(x, 3, 2, 1)
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return self.fc(x.mean(dim=(2, 3)))


class _VitBlock(nn.Module):
    def __init__(self, D, heads, mlp):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(D, 3 * D); self.attn.proj = nn.Linear(D, D)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(D, mlp); self.mlp.fc2 = nn.Linear(mlp, D)
        self.heads = heads

    def forward(self, x):
        b, t, d = x.shape
        qkv = self.attn.qkv(self.norm1(x)).reshape(b, t, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        att = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) * (d // self.heads) ** -0.5, dim=-1)
        x = x + self.attn.proj((att @ qkv[2]).transpose(1, 2).reshape(b, t, d))
        return x + self.mlp.fc2(nn.functional.gelu(self.mlp.fc1(self.norm2(x)), approximate="tanh"))


class VisionTransformer(nn.Module):
    def __init__(self, D=128, depth=2, heads=2, mlp=256, patch=16, in_hw=(64, 64), num_classes=1000):
        super().__init__()
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, D, patch, patch)
        ntok = (in_hw[0] // patch) * (in_hw[1] // patch) + 1
        self.cls_token = nn.Parameter(torch.randn(1, 1, D) * 0.2)
        self.pos_embed = nn.Parameter(torch.randn(1, ntok, D) * 0.2)
        self.blocks = nn.Sequential(*[_VitBlock(D, heads, mlp) for _ in range(depth)])
        self.norm = nn.LayerNorm(D, eps=1e-6)
        self.head = nn.Linear(D, num_classes)

    def forward(self, x):
        e = self.patch_embed.proj(x).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(e.shape[0], -1, -1), e], dim=1) + self.pos_embed
        return self.head(self.norm(self.blocks(x))[:, 0])
