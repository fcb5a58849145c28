"""CPU oracle: ConvNeXt (timm `convnext_tiny` family) in plain torch (TEST INFRASTRUCTURE ONLY).

Block / stem / downsample arithmetic follows the reference tree's own ConvNeXt definition
(/root/reference/semantic_segmentation/backbone/convnext.py:21-56 block, :79-88 stem and downsample, :158-182
LayerNorm) with timm's classification head (global average pool -> LayerNorm -> Linear) and timm's parameter names
[recall]; timm cannot be imported.  Parameter count 28,589,128 for convnext_tiny is
checked in tests/test_oracle_cpu.py.  Stochastic-depth masks are INJECTED (per block, float [B], already scaled by
1/keep_prob) so that parity tests do not depend on RNG streams.  `bf16_points=True`: see oracle/resnet_ref.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet_ref import _RoundBF16, _RoundWeight

CONFIGS = {"convnext_tiny": ((3, 3, 9, 3), (96, 192, 384, 768)), "convnext_small": ((3, 3, 27, 3), (96, 192, 384, 768)),
           "convnext_test": ((1, 1, 2, 1), (32, 64, 128, 192)),
           # the configuration of tests/golden/convnext_ref_vectors.npz (generated from the reference's own backbone class)
           "convnext_pin": ((1, 1, 1, 1), (32, 64, 96, 192))}


def _r(x, on):
    return _RoundBF16.apply(x) if on else x


def _w(w, on):
    return _RoundWeight.apply(w) if on else w


def _ln2d(x, ln):
    return F.layer_norm(x.permute(0, 2, 3, 1), ln.normalized_shape, ln.weight, ln.bias, ln.eps).permute(0, 3, 1, 2)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _Block(nn.Module):
    def __init__(self, dim, q):
        super().__init__()
        self.q = q
        self.conv_dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)
        self.gamma = nn.Parameter(torch.full((dim,), 1e-6))
        self.keep = None   # injected per-sample stochastic-depth factor (float [B]) or None

    def forward(self, x):
        q = self.q
        d = _r(F.conv2d(x, _w(self.conv_dw.weight, q), self.conv_dw.bias, padding=3, groups=x.shape[1]), q)
        h = _r(self.norm(d.permute(0, 2, 3, 1)), q)
        z1 = _r(F.linear(h, _w(self.mlp.fc1.weight, q), self.mlp.fc1.bias), q)
        a = _r(F.gelu(z1), q)
        z2 = _r(F.linear(a, _w(self.mlp.fc2.weight, q), self.mlp.fc2.bias), q)
        k = 1.0 if self.keep is None else self.keep.to(x.dtype).view(-1, 1, 1, 1)
        return _r(x + (k * self.gamma * z2).permute(0, 3, 1, 2), q)


class _Stage(nn.Module):
    def __init__(self, prev, dim, depth, q, first):
        super().__init__()
        self.downsample = nn.Identity() if first else nn.Sequential(nn.LayerNorm(prev, eps=1e-6), nn.Conv2d(prev, dim, 2, 2))
        self.blocks = nn.Sequential(*[_Block(dim, q) for _ in range(depth)])


class _Head(nn.Module):
    def __init__(self, dim, num_classes):
        super().__init__()
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.fc = nn.Linear(dim, num_classes)


class ConvNeXtRef(nn.Module):
    def __init__(self, arch="convnext_tiny", num_classes=1000, bf16_points=False):
        super().__init__()
        depths, dims = CONFIGS[arch]
        self.q = bf16_points
        self.stem = nn.Sequential(nn.Conv2d(3, dims[0], 4, 4), nn.LayerNorm(dims[0], eps=1e-6))
        self.stages = nn.Sequential(*[_Stage(dims[i - 1] if i else dims[0], dims[i], depths[i], bf16_points, i == 0)
                                      for i in range(4)])
        self.head = _Head(dims[-1], num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def all_blocks(self):
        return [b for st in self.stages for b in st.blocks]

    def forward_features(self, x):
        """Raw output of every stage (reference backbone: convnext.py:138-150 before its per-output norms)."""
        q = self.q
        x = _r(x, q)
        s = _r(F.conv2d(x, _w(self.stem[0].weight, q), self.stem[0].bias, stride=4), q)
        x = _r(_ln2d(s, self.stem[1]), q)
        feats = []
        for st in self.stages:
            if not isinstance(st.downsample, nn.Identity):
                ln = _r(_ln2d(x, st.downsample[0]), q)
                x = _r(F.conv2d(ln, _w(st.downsample[1].weight, q), st.downsample[1].bias, stride=2), q)
            x = st.blocks(x)
            feats.append(x)
        return feats

    def forward(self, x):
        q = self.q
        x = self.forward_features(x)[-1]
        pool = _r(x.mean(dim=(2, 3)), q)
        pn = _r(self.head.norm(pool), q)
        return _r(F.linear(pn, _w(self.head.fc.weight, q), self.head.fc.bias), q)
