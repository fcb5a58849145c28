"""CPU oracle: ResNet-18/34/50 in plain torch (TEST INFRASTRUCTURE ONLY -- never imported by the product).

The reference has no model code: it calls timm.create_model(args.model) (/root/reference/train.py:194) and timm is
neither in the reference tree nor installed here, so this file restates the architecture timm's `resnet18/34/50`
implement (ResNet v1.5: 7x7/2 stem, 3x3/2 max-pool, [Basic|Bottleneck] blocks with the stride on the 3x3,
1x1/stride projection shortcut + BN, global average pool, linear head; BatchNorm eps 1e-5 momentum 0.1;
Kaiming-normal fan_out conv init, zero-init of each block's last BN weight) with timm's parameter names.
Pinned against torch itself only (parameter counts 11,689,512 / 25,557,032 from SURVEY.md Appendix A are
checked in tests/test_oracle_cpu.py): "parity unpinned" with respect to timm, which cannot be imported.

`bf16_points=True` inserts the rounding points of the HIP path (inputs, filters, every stored activation and
every stored activation-gradient rounded to bf16; accumulation, BatchNorm statistics and weight gradients in
fp32), so that logits/gradients can be compared at the 1e-3 level instead of bf16's 4e-3 per-op noise.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F


class _RoundBF16(torch.autograd.Function):
    """Round to the bf16 grid, keeping the working dtype (fp32, or fp64 for the re-association calibration run)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundWeight(torch.autograd.Function):
    """bf16 filter copy of an fp32 master weight; the gradient stays fp32 (straight through)."""

    @staticmethod
    def forward(ctx, w):
        return w.to(torch.bfloat16).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _r(x, on):
    return _RoundBF16.apply(x) if on else x


def _w(w, on):
    return _RoundWeight.apply(w) if on else w


class _Block(nn.Module):
    def __init__(self, kind, inplanes, planes, stride, bf16_points):
        super().__init__()
        self.kind, self.q = kind, bf16_points
        self.trace, self.trace_name = None, ""
        exp = 4 if kind == "bottleneck" else 1
        out = planes * exp
        if kind == "bottleneck":
            self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, out, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(out)
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != out:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride, bias=False), nn.BatchNorm2d(out))

    def _cb(self, conv, bn, x, tag=None):
        y = _r(F.conv2d(x, _w(conv.weight, self.q), None, conv.stride, conv.padding), self.q)
        if self.trace is not None and tag is not None:
            self.trace[f"{self.trace_name}.{tag}.y"] = y.detach()
        return bn(y)

    def _keep(self, tag, t):
        if self.trace is not None:
            self.trace[f"{self.trace_name}.{tag}"] = t.detach()
        return t

    def forward(self, x):
        q = self.q
        idn = x
        if self.downsample is not None:
            idn = self._keep("down.a", _r(self._cb(self.downsample[0], self.downsample[1], x, "down"), q))
        if self.kind == "bottleneck":
            o = self._keep("0.a", _r(F.relu(self._cb(self.conv1, self.bn1, x, "0")), q))
            o = self._keep("1.a", _r(F.relu(self._cb(self.conv2, self.bn2, o, "1")), q))
            o = self._cb(self.conv3, self.bn3, o, "2")
            last = "2.a"
        else:
            o = self._keep("0.a", _r(F.relu(self._cb(self.conv1, self.bn1, x, "0")), q))
            o = self._cb(self.conv2, self.bn2, o, "1")
            last = "1.a"
        return self._keep(last, _r(F.relu(o + idn), q))

    @property
    def last_bn(self):
        return self.bn3 if self.kind == "bottleneck" else self.bn2


ARCHS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]), "resnet50": ("bottleneck", [3, 4, 6, 3])}


class ResNetRef(nn.Module):
    def __init__(self, arch="resnet50", num_classes=1000, bf16_points=False, zero_init_last=True):
        super().__init__()
        kind, layers = ARCHS[arch]
        self.q = bf16_points
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        exp = 4 if kind == "bottleneck" else 1
        for li, (planes, n) in enumerate(zip([64, 128, 256, 512], layers)):
            blocks = []
            for bi in range(n):
                blocks.append(_Block(kind, inplanes, planes, 2 if (bi == 0 and li > 0) else 1, bf16_points))
                inplanes = planes * exp
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.fc = nn.Linear(inplanes, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if zero_init_last:
            for m in self.modules():
                if isinstance(m, _Block):
                    nn.init.zeros_(m.last_bn.weight)

    def set_trace(self, trace):
        """trace: dict that receives every stored intermediate (NCHW), keyed like the HIP workspace."""
        self.trace = trace
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                blk.trace, blk.trace_name = trace, f"layer{li}.{bi}"

    def forward(self, x):
        q = self.q
        tr = getattr(self, "trace", None)
        x = _r(x, q)
        y = _r(F.conv2d(x, _w(self.conv1.weight, q), None, 2, 3), q)
        a0 = _r(F.relu(self.bn1(y)), q)
        p0 = F.max_pool2d(a0, 3, 2, 1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(p0))))
        pooled = _r(x.mean(dim=(2, 3)), q)
        logits = _r(F.linear(pooled, _w(self.fc.weight, q), self.fc.bias), q)
        if tr is not None:
            tr.update({"y0": y.detach(), "a0": a0.detach(), "p0": p0.detach(), "pooled": pooled.detach(),
                       "logits": logits.detach()})
        return logits
