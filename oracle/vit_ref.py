"""CPU oracle: ViT (timm `vit_base_patch16_224` family) in plain torch (TEST INFRASTRUCTURE ONLY).

timm is absent: the architecture is restated from its published behaviour [recall] -- patch-embedding
conv (k = s = 16, bias), class token, learned position embedding, pre-LayerNorm blocks (eps 1e-6) with fused-QKV
multi-head attention (scale head_dim^-0.5) and a 4x exact-GELU MLP, final LayerNorm, class-token pooling, linear
head -- with timm's parameter names.  Pinned against torch only (ViT-B/16 parameter count 86,567,656 from SURVEY
Appendix A.3 is checked in tests/test_oracle_cpu.py): "parity unpinned" with respect to timm.
`bf16_points=True` inserts the HIP path's rounding points (see oracle/resnet_ref.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet_ref import _RoundBF16, _RoundWeight

CONFIGS = {"vit_base_patch16_224": (16, 768, 12, 12, 4), "vit_small_patch16_224": (16, 384, 12, 6, 4),
           "vit_tiny_test": (16, 128, 2, 2, 4)}


def _r(x, on):
    return _RoundBF16.apply(x) if on else x


def _w(w, on):
    return _RoundWeight.apply(w) if on else w


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, heads, hidden, q):
        super().__init__()
        self.q = q
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, hidden)

    def forward(self, x):
        q = self.q
        B, T, D = x.shape
        H = self.attn.heads
        h = _r(self.norm1(x), q)
        qkv = _r(F.linear(h, _w(self.attn.qkv.weight, q), self.attn.qkv.bias), q)
        qq, kk, vv = qkv.reshape(B, T, 3, H, D // H).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qq @ kk.transpose(-1, -2)) * (D // H) ** -0.5, dim=-1)
        ao = _r((att @ vv).transpose(1, 2).reshape(B, T, D), q)
        x1 = _r(x + F.linear(ao, _w(self.attn.proj.weight, q), self.attn.proj.bias), q)
        h2 = _r(self.norm2(x1), q)
        z = _r(F.linear(h2, _w(self.mlp.fc1.weight, q), self.mlp.fc1.bias), q)
        a = _r(F.gelu(z), q)
        return _r(x1 + F.linear(a, _w(self.mlp.fc2.weight, q), self.mlp.fc2.bias), q)


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)


class ViTRef(nn.Module):
    def __init__(self, arch="vit_base_patch16_224", num_classes=1000, img_size=224, bf16_points=False):
        super().__init__()
        patch, dim, depth, heads, ratio = CONFIGS[arch]
        self.q = bf16_points
        self.patch_embed = _PatchEmbed(patch, dim)
        n = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.blocks = nn.Sequential(*[_Block(dim, heads, dim * ratio, bf16_points) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Linear(dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        q = self.q
        x = _r(x, q)
        p = _r(F.conv2d(x, _w(self.patch_embed.proj.weight, q), self.patch_embed.proj.bias, self.patch_embed.proj.stride), q)
        p = p.flatten(2).transpose(1, 2)
        B = p.shape[0]
        tok = torch.cat([self.cls_token.expand(B, -1, -1), p], dim=1)
        x = _r(tok + self.pos_embed, q)
        x = self.blocks(x)
        pooled = _r(self.norm(x[:, 0]), q)
        return _r(F.linear(pooled, _w(self.head.weight, q), self.head.bias), q)
