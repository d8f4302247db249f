"""Minimal torch definitions with the naming of the usual trained checkpoints (torchvision ResNet, timm
VisionTransformer) - test infrastructure for weights.from_state_dict; torchvision and timm are not installed here."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from oracle.torch_fp32 import ResNet, _Basic, _Bottleneck  # noqa: E402,F401  (the fp32 ResNet lives beside the other CPU checkers)


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
        return x + self.mlp.fc2(nn.functional.gelu(self.mlp.fc1(self.norm2(x))))


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
