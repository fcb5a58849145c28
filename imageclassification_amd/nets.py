"""ResNet-18/34/50 on the gfx950 kernels: static layer lists with hand-written forward and backward.

The reference builds its network with timm.create_model (/root/reference/train.py:194) and runs it through
autograd (engine.py:48,51,64,72); there is no model code in the reference tree.  Here a model is a flat list
of layer records over four flat arenas (fp32 parameters, fp32 gradients, bf16 "shadow" filters the conv
kernels read, fp32 BatchNorm buffers) and every layer's forward/backward is one or more C-ABI calls
(include/icamd.h).  No autograd graph, no torch ops on the step path.

Architecture = timm/torchvision ResNet v1.5 (stride on the 3x3 of a bottleneck), BatchNorm eps 1e-5,
momentum 0.1, parameter names identical to timm's (`conv1.weight`, `layer1.0.bn1.weight`, ..., `fc.bias`) so
state_dicts interchange with the CPU oracle and with timm checkpoints.

Data layout in HBM: activations NHWC bf16; filters [Cout][KH][KW][Cin] (stem Cin zero-padded 3->8, FC rows
zero-padded to a multiple of 64); everything 256 B aligned inside the arenas.
"""
import ctypes
import math
import os
from collections import OrderedDict

import torch

from . import hip
from .checkpoint import PicklableModel

BN_EPS = 1e-5
_FUSED_BNBWD = os.environ.get("ICAMD_FUSED_BNBWD", "0") == "1"
_WGRAD_STREAM = os.environ.get("ICAMD_WGRAD_STREAM", "1") != "0"
_DUAL_BNBWD = os.environ.get("ICAMD_DUAL_BNBWD", "1") != "0"
_SUB2_SHORTCUT = os.environ.get("ICAMD_SUB2_SHORTCUT", "1") != "0"
# stem max-pool backward folded into the BatchNorm backward (icamd_bn_bwd_maxpool3x3s2, bit-identical): measured on MI355X it
# trades 0.20 ms of pooling backward for +0.27 ms of BatchNorm backward (the gather of <= 4 windows per pixel, done in both
# passes, is bound by L1 / texture-address requests, not by the HBM bytes it saves), so it is opt-in
_FUSED_POOL_BWD = os.environ.get("ICAMD_FUSED_POOL_BWD", "0") == "1"
BN_MOMENTUM = 0.1

ARCHS = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
}


def _align(n, a):
    return (n + a - 1) // a * a


class _Param:
    """One logical parameter: a slice of the flat arenas plus its torch-layout shape."""
    __slots__ = ("name", "offset", "numel", "torch_shape", "kind", "padded_shape")

    def __init__(self, name, offset, numel, torch_shape, kind, padded_shape):
        self.name, self.offset, self.numel = name, offset, numel
        self.torch_shape, self.kind, self.padded_shape = torch_shape, kind, padded_shape


class _Conv:
    """Convolution record. weight param in arena layout [Cout_p][KH][KW][Cin_p]."""

    def __init__(self, name, cin, cout, k, stride, pad, cin_p=None, cout_p=None, bias=False, k_p=None):
        self.name = name
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.cin_p = cin_p or cin
        self.cout_p = cout_p or cout
        self.k_p = k_p or k        # kernel extent of the arena layout (the stem stores 7x7 filters as 8x8x4, see _build_graph)
        self.has_bias = bias
        self.w = None      # _Param
        self.b = None
        self.wt_offset = None  # offset into the transposed shadow arena (None: no data gradient needed)
        self.descs = {}    # (N, IH, IW) -> ConvDesc

    def desc(self, N, IH, IW):
        key = (N, IH, IW)
        d = self.descs.get(key)
        if d is None:
            d = hip.conv_desc(N, IH, IW, self.cin_p, self.cout_p, self.k, self.k, self.stride, self.pad)
            self.descs[key] = d
        return d


class _BN:
    def __init__(self, name, c):
        self.name, self.c = name, c
        self.weight = self.bias = None  # _Param
        self.buf_offset = None          # running_mean at buf_offset, running_var at buf_offset + c
        self.stat_offset = None         # mean, invstd, scale, shift in the per-model stat arena (4*c floats)


class ResNet(PicklableModel):
    """HIP ResNet. `model(x)` runs the forward and returns bf16 logits [B, num_classes] (a view)."""

    def __init__(self, arch="resnet50", num_classes=1000, device="cuda", zero_init_last=True, seed=None):
        hip.require_gpu()
        self.lib = hip.load()
        self.arch = arch
        self.num_classes = num_classes
        self.device = torch.device(device)
        self.training = True
        self._fold_dirty = True
        # eval forwards use BatchNorm-folded filters (ICAMD_EVAL_FOLD=0 keeps the separate BatchNorm pass)
        self.fold_eval = os.environ.get("ICAMD_EVAL_FOLD", "1") != "0"
        block, layers = ARCHS[arch]
        self.block = block
        self.expansion = 4 if block == "bottleneck" else 1
        self.ncls_p = _align(num_classes, 64)
        self._build_graph(layers)
        self._allocate()
        self.num_batches_tracked = 0
        self._ws = {}
        self._streams_ready = False
        self.wgrad_side_stream = True   # bench.py turns this off for its per-kernel timing pass
        self.grad_ready_hook = None   # called with (param_offset_lo, param_offset_hi) as gradients complete
        self.init_weights(zero_init_last=zero_init_last, seed=seed)

    # ------------------------------------------------------------------ structure
    def _build_graph(self, layers):
        self.convs, self.bns = [], []
        # stem filters live as [64][8][8][4] (row 7, column 7, channel 3 zero): the layout icamd_stem7x7s2_fwd / _wgrad
        # reduce over, on the [N][H][W+8][4] image icamd_pack_input_rgb4 writes
        self.stem_conv = self._conv("conv1", 3, 64, 7, 2, 3, cin_p=4, k_p=8)
        self.stem_bn = self._bn("bn1", 64)
        self.blocks = []
        inplanes = 64
        for li, (planes, nblocks) in enumerate(zip([64, 128, 256, 512], layers)):
            for bi in range(nblocks):
                stride = 2 if (bi == 0 and li > 0) else 1
                name = f"layer{li + 1}.{bi}"
                outplanes = planes * self.expansion
                blk = {"name": name, "stride": stride}
                if self.block == "bottleneck":
                    blk["convs"] = [self._conv(f"{name}.conv1", inplanes, planes, 1, 1, 0),
                                    self._conv(f"{name}.conv2", planes, planes, 3, stride, 1),
                                    self._conv(f"{name}.conv3", planes, outplanes, 1, 1, 0)]
                    blk["bns"] = [self._bn(f"{name}.bn1", planes), self._bn(f"{name}.bn2", planes),
                                  self._bn(f"{name}.bn3", outplanes)]
                else:
                    blk["convs"] = [self._conv(f"{name}.conv1", inplanes, planes, 3, stride, 1),
                                    self._conv(f"{name}.conv2", planes, planes, 3, 1, 1)]
                    blk["bns"] = [self._bn(f"{name}.bn1", planes), self._bn(f"{name}.bn2", planes)]
                if stride != 1 or inplanes != outplanes:
                    blk["down_conv"] = self._conv(f"{name}.downsample.0", inplanes, outplanes, 1, stride, 0)
                    blk["down_bn"] = self._bn(f"{name}.downsample.1", outplanes)
                self.blocks.append(blk)
                inplanes = outplanes
        self.feat_dim = inplanes
        self.fc = self._conv("fc", inplanes, self.num_classes, 1, 1, 0, cout_p=self.ncls_p, bias=True)

    def _conv(self, name, cin, cout, k, stride, pad, cin_p=None, cout_p=None, bias=False, k_p=None):
        c = _Conv(name, cin, cout, k, stride, pad, cin_p, cout_p, bias, k_p)
        self.convs.append(c)
        return c

    def _bn(self, name, c):
        b = _BN(name, c)
        self.bns.append(b)
        return b

    def _allocate(self):
        dev = self.device
        # parameter order = timm/torchvision module order (conv, bn, ..., downsample, fc)
        order = [self.stem_conv, self.stem_bn]
        for blk in self.blocks:
            for c, b in zip(blk["convs"], blk["bns"]):
                order += [c, b]
            if "down_conv" in blk:
                order += [blk["down_conv"], blk["down_bn"]]
        order.append(self.fc)
        self.params = OrderedDict()
        off = 0

        def add(name, torch_shape, kind, padded_shape):
            nonlocal off
            numel = 1
            for s in padded_shape:
                numel *= s
            p = _Param(name, off, numel, tuple(torch_shape), kind, tuple(padded_shape))
            self.params[name] = p
            off = _align(off + numel, 64)  # 256 B alignment of every fp32 slice
            return p

        boff = 0
        soff = 0
        for m in order:
            if isinstance(m, _Conv):
                wname = m.name + ".weight"
                m.w = add(wname, (m.cout, m.cin, m.k, m.k), "conv", (m.cout_p, m.k_p, m.k_p, m.cin_p))
                if m.has_bias:
                    m.b = add(m.name + ".bias", (m.cout,), "vec", (m.cout_p,))
            else:
                m.weight = add(m.name + ".weight", (m.c,), "vec", (m.c,))
                m.bias = add(m.name + ".bias", (m.c,), "vec", (m.c,))
                m.buf_offset = boff
                boff = _align(boff + 2 * m.c, 64)
                m.stat_offset = soff
                soff = _align(soff + 4 * m.c, 64)
        self.n_params = off
        self.param_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.buffer_arena = torch.zeros(max(boff, 64), dtype=torch.float32, device=dev)
        self.stat_arena = torch.zeros(max(soff, 64), dtype=torch.float32, device=dev)
        # transposed filters for the data-gradient kernels (every conv except the stem)
        toff = 0
        descs, jobs, tjobs = [], [], []
        for m in self.convs:
            if m is self.stem_conv:
                continue
            m.wt_offset = toff
            T = m.k * m.k
            descs.append([m.w.offset, toff, m.cout_p, T, m.cin_p, 0, 0, 0])
            if m.cout_p % 64 == 0 and m.cin_p % 64 == 0:
                for t in range(T):
                    for co0 in range(0, m.cout_p, 64):
                        for ci0 in range(0, m.cin_p, 64):
                            tjobs.append([len(descs) - 1, t, co0, ci0])
            else:
                for s in range(0, m.w.numel, 4096):
                    jobs.append([len(descs) - 1, s])
            toff = _align(toff + m.w.numel, 128)
        self.shadow_t = torch.zeros(toff, dtype=torch.bfloat16, device=dev)
        self._tr_descs = torch.tensor(descs, dtype=torch.int64, device=dev)
        self._tr_jobs = torch.tensor(jobs if jobs else [[0, 0]], dtype=torch.int32, device=dev)
        self._tr_njobs = len(jobs)
        self._tr_tjobs = torch.tensor(tjobs if tjobs else [[0, 0, 0, 0]], dtype=torch.int32, device=dev)
        self._tr_ntjobs = len(tjobs)

    # ------------------------------------------------------------------ parameters / state_dict
    def _ctor_kwargs(self):
        return {"arch": self.arch, "num_classes": self.num_classes}

    def init_weights(self, zero_init_last=True, seed=None):
        """timm ResNet.init_weights: Kaiming-normal (fan_out, relu) convs, BN weight 1 / bias 0, zero-init of the
        last BN weight of each residual block, nn.Linear default init for the classifier."""
        g = torch.Generator()
        if seed is not None:
            g.manual_seed(seed)
        else:
            g.manual_seed(torch.initial_seed() % (2 ** 63))
        sd = OrderedDict()
        for name, p in self.params.items():
            if p.kind == "conv" and not name.startswith("fc."):
                cout, cin, k, _ = p.torch_shape
                std = math.sqrt(2.0 / (cout * k * k))
                sd[name] = torch.randn(p.torch_shape, generator=g) * std
            elif name == "fc.weight":
                bound = 1.0 / math.sqrt(p.torch_shape[1])
                sd[name] = (torch.rand(p.torch_shape[0], p.torch_shape[1], generator=g) * 2 - 1) * bound
            elif name == "fc.bias":
                bound = 1.0 / math.sqrt(self.feat_dim)
                sd[name] = (torch.rand(p.torch_shape, generator=g) * 2 - 1) * bound
            elif name.endswith(".weight"):
                sd[name] = torch.ones(p.torch_shape)
            else:
                sd[name] = torch.zeros(p.torch_shape)
        if zero_init_last:
            for blk in self.blocks:
                sd[blk["bns"][-1].name + ".weight"].zero_()
        for b in self.bns:
            sd[b.name + ".running_mean"] = torch.zeros(b.c)
            sd[b.name + ".running_var"] = torch.ones(b.c)
            sd[b.name + ".num_batches_tracked"] = torch.tensor(0)
        self.load_state_dict(sd)

    def _to_arena_layout(self, p, t):
        t = t.detach().to(torch.float32).cpu()
        if p.kind == "conv":
            if t.dim() == 2:
                t = t[:, :, None, None]
            cout, cin, kh, kw = t.shape
            full = torch.zeros(p.padded_shape)
            full[:cout, :kh, :kw, :cin] = t.permute(0, 2, 3, 1)
            return full.flatten()
        full = torch.zeros(p.padded_shape)
        full[: t.numel()] = t.flatten()
        return full

    def _from_arena_layout(self, p, flat):
        t = flat.reshape(p.padded_shape)
        if p.kind == "conv":
            cout, cin, kh, kw = (tuple(p.torch_shape) + (1, 1))[:4]
            t = t[:cout, :kh, :kw, :cin].permute(0, 3, 1, 2).contiguous()
            if p.name == "fc.weight":
                t = t.reshape(cout, cin)
            return t
        return t[: p.torch_shape[0]].clone()

    def load_state_dict(self, sd, strict=True):
        host = self.param_arena.cpu()
        missing = []
        for name, p in self.params.items():
            if name not in sd:
                missing.append(name)
                continue
            host[p.offset:p.offset + p.numel] = self._to_arena_layout(p, sd[name])
        bufs = self.buffer_arena.cpu()
        for b in self.bns:
            for j, key in enumerate(("running_mean", "running_var")):
                k = f"{b.name}.{key}"
                if k in sd:
                    bufs[b.buf_offset + j * b.c: b.buf_offset + (j + 1) * b.c] = sd[k].detach().float().cpu()
                else:
                    missing.append(k)
        k = f"{self.bns[0].name}.num_batches_tracked"
        if k in sd:
            self.num_batches_tracked = int(sd[k])
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self.param_arena.copy_(host)
        self.buffer_arena.copy_(bufs)
        self.refresh_shadow()
        self._fold_dirty = True
        return missing

    def state_dict(self):
        host = self.param_arena.cpu()
        bufs = self.buffer_arena.cpu()
        sd = OrderedDict()
        by_module = OrderedDict()
        for name, p in self.params.items():
            by_module.setdefault(name.rsplit(".", 1)[0], []).append(p)
        bn_by_name = {b.name: b for b in self.bns}
        for mod, plist in by_module.items():
            for p in plist:
                sd[p.name] = self._from_arena_layout(p, host[p.offset:p.offset + p.numel])
            if mod in bn_by_name:
                b = bn_by_name[mod]
                sd[mod + ".running_mean"] = bufs[b.buf_offset:b.buf_offset + b.c].clone()
                sd[mod + ".running_var"] = bufs[b.buf_offset + b.c:b.buf_offset + 2 * b.c].clone()
                sd[mod + ".num_batches_tracked"] = torch.tensor(self.num_batches_tracked)
        return sd

    def named_parameters(self):
        """(name, fp32 arena view) pairs; the views alias the flat parameter arena."""
        for name, p in self.params.items():
            yield name, self.param_arena[p.offset:p.offset + p.numel]

    def parameters(self):
        for _, v in self.named_parameters():
            yield v

    def grad_of(self, name):
        """Gradient of a parameter in torch layout (host copy), for tests and checkpoint tools."""
        p = self.params[name]
        return self._from_arena_layout(p, self.grad_arena[p.offset:p.offset + p.numel].cpu())

    def refresh_shadow(self):
        """Re-derive the bf16 filters (and their transposes) from the fp32 master parameters."""
        s = hip.stream_ptr()
        hip.check(self.lib.icamd_f32_to_bf16(self.param_arena.data_ptr(), self.shadow.data_ptr(), self.n_params, s),
                  "f32_to_bf16")
        self.refresh_transposed()

    def refresh_transposed(self):
        s = hip.stream_ptr()
        if self._tr_ntjobs:
            hip.check(self.lib.icamd_filter_transpose_tiled(self.shadow.data_ptr(), self.shadow_t.data_ptr(),
                                                            self._tr_descs.data_ptr(), self._tr_tjobs.data_ptr(),
                                                            self._tr_ntjobs, s), "filter_transpose_tiled")
        if self._tr_njobs:
            hip.check(self.lib.icamd_filter_transpose(self.shadow.data_ptr(), self.shadow_t.data_ptr(),
                                                      self._tr_descs.data_ptr(), self._tr_jobs.data_ptr(), self._tr_njobs,
                                                      s), "filter_transpose")

    def train(self, mode=True):
        self.training = bool(mode)
        self._fold_dirty = True     # parameters / running statistics may move before the next eval forward
        return self

    # ------------------------------------------------------------------ inference form
    def fold_batchnorm(self):
        """Eval fast path: fold every BatchNorm (running statistics) into the bf16 filters of the convolution in front
        of it, so an eval forward is one kernel per convolution (bias = BN shift, residual add and ReLU in the epilogue)
        with no BatchNorm pass over the activations.  Re-done whenever the mode, the weights or the EMA changed."""
        dev = self.device
        if getattr(self, "shadow_eval", None) is None:
            self.shadow_eval = torch.empty_like(self.shadow)
            self.eval_shift = torch.zeros(sum(b.c for b in self.bns), dtype=torch.float32, device=dev)
            off = 0
            for b in self.bns:
                b.shift_offset = off
                off += b.c
        s = hip.stream_ptr()
        pairs = [(self.stem_conv, self.stem_bn)]
        for blk in self.blocks:
            pairs += list(zip(blk["convs"], blk["bns"]))
            if "down_conv" in blk:
                pairs.append((blk["down_conv"], blk["down_bn"]))
        for conv, bn in pairs:
            rm = self.buffer_arena.data_ptr() + 4 * bn.buf_offset
            hip.check(self.lib.icamd_bn_fold_filters(self._pf(conv.w), self._pf(bn.weight), self._pf(bn.bias), rm,
                                                     rm + 4 * bn.c, BN_EPS, conv.cout_p, conv.w.numel // conv.cout_p,
                                                     self.shadow_eval.data_ptr() + 2 * conv.w.offset,
                                                     self.eval_shift.data_ptr() + 4 * bn.shift_offset, s), bn.name)
        self._fold_dirty = False

    def _conv_act_eval(self, conv, bn, x, N, IH, IW, out, residual, relu, s):
        d = conv.desc(N, IH, IW)
        if conv is self.stem_conv:
            hip.check(self.lib.icamd_stem7x7s2_fwd(x, self.shadow_eval.data_ptr() + 2 * conv.w.offset, out.data_ptr(),
                                                   self.eval_shift.data_ptr() + 4 * bn.shift_offset, None, int(relu), N, IH, IW,
                                                   conv.cout_p, s), conv.name)
            return d
        hip.check(self.lib.icamd_conv2d_fwd_act(ctypes.byref(d), x, self.shadow_eval.data_ptr() + 2 * conv.w.offset,
                                                out.data_ptr(), self.eval_shift.data_ptr() + 4 * bn.shift_offset, residual,
                                                int(relu), s), conv.name)
        return d

    def _forward_eval_folded(self, ws):
        lib = self.lib
        s = hip.stream_ptr()
        N, H, W = ws["N"], ws["H"], ws["Ws"]
        if self._fold_dirty:
            self.fold_batchnorm()
        d0 = self._conv_act_eval(self.stem_conv, self.stem_bn, ws["x8"].data_ptr(), N, H, W, ws["a0"], None, True, s)
        hip.check(lib.icamd_maxpool3x3s2_fwd(ws["a0"].data_ptr(), ws["p0"].data_ptr(), None, N, d0.OH, d0.OW, 64, s),
                  "maxpool")
        x = ws["p0"]
        h, w = x.shape[1], x.shape[2]
        for blk, b in zip(self.blocks, ws["blocks"]):
            if "down_conv" in blk:
                self._conv_act_eval(blk["down_conv"], blk["down_bn"], x.data_ptr(), N, h, w, b["ad"], None, False, s)
                idn = b["ad"]
            else:
                idn = x
            cur, ch, cw = x, h, w
            n = len(blk["convs"])
            for i, (conv, bn) in enumerate(zip(blk["convs"], blk["bns"])):
                d = self._conv_act_eval(conv, bn, cur.data_ptr(), N, ch, cw, b["a"][i],
                                        idn.data_ptr() if i == n - 1 else None, True, s)
                cur, ch, cw = b["a"][i], d.OH, d.OW
            x, h, w = cur, ch, cw
        hip.check(lib.icamd_avgpool_fwd(x.data_ptr(), ws["pooled"].data_ptr(), N, h * w, self.feat_dim, s), "avgpool")
        dfc = self.fc.desc(N, 1, 1)
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(dfc), ws["pooled"].data_ptr(), self._w(self.fc),
                                       ws["logits"].data_ptr(), self._pf(self.fc.b), None, None, s), "fc")
        return ws["logits"]

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, N, H, W):
        key = (N, H, W)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev = self.device
        lib = self.lib
        ws = {"N": N, "H": H, "W": W}

        def act(n, h, w, c):
            return torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)

        # the rgb4 stem layout (3 + 5 zero columns per row); an odd width gets one more zero column -- part of the
        # convolution's own padding -- and the stem kernels run on the even width Ws
        Ws = W + (W & 1)
        ws["Ws"] = Ws
        ws["x8"] = act(N, H, Ws + 8, 4)
        d0 = self.stem_conv.desc(N, H, Ws)
        ws["y0"] = act(N, d0.OH, d0.OW, 64)
        ws["a0"] = act(N, d0.OH, d0.OW, 64)
        PH, PW = (d0.OH - 1) // 2 + 1, (d0.OW - 1) // 2 + 1
        ws["p0"] = act(N, PH, PW, 64)
        ws["p0_idx"] = torch.empty(N, PH, PW, 64, dtype=torch.uint8, device=dev)
        max_act = max(ws["y0"].numel(), ws["x8"].numel())
        max_stats = lib.icamd_conv2d_stats_rows(ctypes.byref(d0)) * 2 * 64
        max_wg = lib.icamd_stem7x7s2_wgrad_workspace_bytes(N, H, Ws, 64)
        max_bnb = lib.icamd_bn_bwd_workspace_bytes(N * d0.OH * d0.OW, 64)
        h, w = PH, PW
        blocks_ws = []
        for blk in self.blocks:
            b = {}
            ih, iw = h, w
            ys, acts = [], []
            for ci, conv in enumerate(blk["convs"]):
                d = conv.desc(N, ih, iw)
                ys.append(act(N, d.OH, d.OW, conv.cout_p))
                acts.append(act(N, d.OH, d.OW, conv.cout_p))
                max_act = max(max_act, ys[-1].numel())
                max_stats = max(max_stats, lib.icamd_conv2d_stats_rows(ctypes.byref(d)) * 2 * conv.cout_p)
                max_wg = max(max_wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d)))
                max_bnb = max(max_bnb, lib.icamd_bn_bwd_workspace_bytes(N * d.OH * d.OW, conv.cout_p))
                ih, iw = d.OH, d.OW
            b["y"], b["a"] = ys, acts
            b["mask"] = torch.empty(acts[-1].numel() // 8, dtype=torch.uint8, device=dev)   # ReLU mask of the block output
            if "down_conv" in blk:
                dd = blk["down_conv"].desc(N, h, w)
                b["yd"] = act(N, dd.OH, dd.OW, blk["down_conv"].cout_p)
                b["ad"] = act(N, dd.OH, dd.OW, blk["down_conv"].cout_p)
                max_stats = max(max_stats, lib.icamd_conv2d_stats_rows(ctypes.byref(dd)) * 2 * blk["down_conv"].cout_p)
                max_wg = max(max_wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(dd)))
                max_bnb = max(max_bnb, lib.icamd_bn_bwd_workspace_bytes(N * dd.OH * dd.OW, blk["down_conv"].cout_p))
            blocks_ws.append(b)
            h, w = ih, iw
        ws["blocks"] = blocks_ws
        ws["final_hw"] = (h, w)
        ws["pooled"] = torch.empty(N, self.feat_dim, dtype=torch.bfloat16, device=dev)
        ws["logits"] = torch.zeros(N, self.ncls_p, dtype=torch.bfloat16, device=dev)
        dfc = self.fc.desc(N, 1, 1)
        max_wg = max(max_wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(dfc)))
        ws["stats"] = torch.empty(max_stats, dtype=torch.float32, device=dev)
        ws["bn_ws"] = torch.zeros(lib.icamd_bn_workspace_bytes(2048), dtype=torch.uint8, device=dev)
        ws["wgrad_ws"] = torch.empty(max_wg, dtype=torch.uint8, device=dev)
        ws["wgrad_ws_bytes"] = max_wg
        ws["bnb_part"] = torch.empty(max_stats + 4 * 2 * 2048, dtype=torch.float32, device=dev)
        ws["bna_ws_bytes"] = lib.icamd_bn_bwd_apply_workspace_bytes(2048)
        ws["bna_ws"] = torch.zeros(ws["bna_ws_bytes"], dtype=torch.uint8, device=dev)
        ws["bnb_ws"] = torch.zeros(max_bnb, dtype=torch.uint8, device=dev)
        ws["bnb_ws2"] = torch.zeros(max_bnb, dtype=torch.uint8, device=dev)   # second BatchNorm of icamd_bn_bwd_dual
        ws["bnb_ws_bytes"] = max_bnb
        ws["max_act"] = max_act
        # loss / metric scratch
        ws["loss_rows"] = torch.empty(N, dtype=torch.float32, device=dev)
        ws["pred"] = torch.empty(N, dtype=torch.int32, device=dev)
        ws["dlogits"] = torch.zeros(N, self.ncls_p, dtype=torch.bfloat16, device=dev)
        ws["dpooled"] = torch.empty(N, self.feat_dim, dtype=torch.bfloat16, device=dev)
        ws["scale_shift_eval"] = torch.empty(2 * 2048, dtype=torch.float32, device=dev)
        self._ws[key] = ws
        return ws

    def _grad_buffers(self, ws):
        """Activation-sized scratch buffers shared by the whole backward pass (allocated on first use): D0, D1, T, DA and a
        pool of four conv-output-gradient buffers (the fused variant uses the first seven as before)."""
        if "gbuf" not in ws:
            n = ws["max_act"]
            ws["gbuf"] = [torch.empty(n, dtype=torch.bfloat16, device=self.device) for _ in range(8)]
        return ws["gbuf"]

    # ------------------------------------------------------------------ primitive wrappers
    def _w(self, conv):
        return self.shadow.data_ptr() + 2 * conv.w.offset

    def _wt(self, conv):
        return self.shadow_t.data_ptr() + 2 * conv.wt_offset

    def _pf(self, p):  # fp32 param pointer
        return self.param_arena.data_ptr() + 4 * p.offset

    def _gf(self, p):  # fp32 grad pointer
        return self.grad_arena.data_ptr() + 4 * p.offset

    def _conv_bn_fwd(self, ws, conv, bn, x, N, IH, IW, y, out, residual, relu, s, maskbits=None, res_bn=None, conv_launch=None):
        """y = conv(x); out = act(bn(y) (+ residual)). Training: batch statistics from the conv epilogue.
        conv_launch(stats_ptr): replaces the convolution launch (round 5: the fused "previous block's apply + this conv1" kernel)."""
        lib = self.lib
        d = conv.desc(N, IH, IW)
        st = self.stat_arena.data_ptr() + 4 * bn.stat_offset
        c = bn.c
        mean, invstd, scale, shift = st, st + 4 * c, st + 8 * c, st + 12 * c
        stem = conv is self.stem_conv

        def conv_fwd(stats_ptr):
            if conv_launch is not None:
                conv_launch(stats_ptr)
                return
            if stem:
                hip.check(lib.icamd_stem7x7s2_fwd(x, self._w(conv), y.data_ptr(), None, stats_ptr, 0, N, IH, IW, conv.cout_p, s),
                          conv.name)
            else:
                hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), x, self._w(conv), y.data_ptr(), None, None, stats_ptr, s),
                          conv.name)

        if self.training:
            stats = ws["stats"].data_ptr()
            conv_fwd(stats)
            rows = lib.icamd_conv2d_stats_rows(ctypes.byref(d))
            rm = self.buffer_arena.data_ptr() + 4 * bn.buf_offset
            hip.check(lib.icamd_bn_train_finalize(stats, rows, c, float(N * d.OH * d.OW), self._pf(bn.weight),
                                                  self._pf(bn.bias), rm, rm + 4 * c, BN_MOMENTUM, BN_EPS, mean, invstd,
                                                  scale, shift, ws["bn_ws"].data_ptr(), s), bn.name)
        else:
            conv_fwd(None)
            rm = self.buffer_arena.data_ptr() + 4 * bn.buf_offset
            scale = ws["scale_shift_eval"].data_ptr()
            shift = scale + 4 * 2048
            hip.check(lib.icamd_bn_eval_coeffs(c, self._pf(bn.weight), self._pf(bn.bias), rm, rm + 4 * c, BN_EPS, scale,
                                               shift, s), bn.name)
        if out is None:      # the caller fuses the apply into its own kernel (stem: BN + ReLU + max-pool)
            return d, scale, shift
        if res_bn is not None:   # residual = raw shortcut conv output; its BatchNorm is applied inside the same pass
            hip.check(lib.icamd_bn_apply_res_bn(y.data_ptr(), scale, shift, residual, res_bn[0], res_bn[1], out.data_ptr(),
                                                maskbits, y.numel(), c, int(relu), s), bn.name)
            return d
        hip.check(lib.icamd_bn_apply(y.data_ptr(), scale, shift, residual, out.data_ptr(), maskbits, y.numel(), c, int(relu),
                                     s), bn.name)
        return d

    # ------------------------------------------------------------------ forward
    def pack(self, x_nchw, mix=None):
        """fp32 NCHW device tensor -> packed NHWC bf16 (channels zero-padded to 8), with optional mixup/cutmix."""
        N, C, H, W = x_nchw.shape
        ws = self._workspace(N, H, W)
        mode, lam, box = (0, 1.0, (0, 0, 0, 0)) if mix is None else mix
        hip.check(self.lib.icamd_pack_input_rgb4(x_nchw.data_ptr(), ws["x8"].data_ptr(), N, C, H, W, mode, float(lam),
                                                 int(box[0]), int(box[1]), int(box[2]), int(box[3]), hip.stream_ptr()), "pack")
        return ws

    def forward_packed(self, ws, logits_only=False):   # logits_only: accepted for interface parity (BatchNorm needs every conv output)
        lib = self.lib
        s = hip.stream_ptr()
        N, H, W = ws["N"], ws["H"], ws["Ws"]
        if not self.training and self.fold_eval:
            return self._forward_eval_folded(ws)
        if self.training:
            self.num_batches_tracked += 1
        # stem: conv -> BatchNorm + ReLU + max-pool in one pass over the conv output (the 112x112 activation is never
        # stored: backward recomputes the ReLU mask from y0 and the pooling argmax is recorded)
        d0, sc0, sh0 = self._conv_bn_fwd(ws, self.stem_conv, self.stem_bn, ws["x8"].data_ptr(), N, H, W, ws["y0"], None, None,
                                         True, s)
        hip.check(lib.icamd_bn_relu_maxpool3x3s2_fwd(ws["y0"].data_ptr(), sc0, sh0, ws["p0"].data_ptr(),
                                                     ws["p0_idx"].data_ptr() if self.training else None, N, d0.OH, d0.OW, 64,
                                                     s), "stem bn+relu+maxpool")
        x = ws["p0"]
        h, w = x.shape[1], x.shape[2]
        # Round 5: the final BatchNorm apply (+ shortcut + ReLU + mask) of a bottleneck block may be DEFERRED into the next block's
        # first convolution (icamd_bn_apply_conv1x1_fused: the block output is written once and multiplied while it is in LDS instead
        # of being re-read by that convolution); `pending` then holds what the apply needs and x is the buffer it will fill.
        pending = None
        nblocks = len(self.blocks)
        for bi, (blk, b) in enumerate(zip(self.blocks, ws["blocks"])):
            b["in"] = x
            b["in_hw"] = (h, w)
            convs, bns = blk["convs"], blk["bns"]
            first_done = False
            if pending is not None:
                # end of the previous block + this block's conv1 (+ statistics of its bn1) in one launch, then bn1 as usual
                pd = pending
                pending = None
                c0, bn0 = convs[0], bns[0]
                d0 = c0.desc(N, h, w)

                def launch(stats_ptr, pd=pd, c0=c0, d0=d0, b=b):
                    hip.check(lib.icamd_bn_apply_conv1x1_fused(ctypes.byref(d0), pd["y"].data_ptr(), pd["scale"], pd["shift"],
                                                               pd["res"], pd["res_scale"], pd["res_shift"], pd["out"].data_ptr(),
                                                               pd["mask"].data_ptr(), self._w(c0), b["y"][0].data_ptr(), stats_ptr, s),
                              pd["name"] + " apply + " + c0.name)

                self._conv_bn_fwd(ws, c0, bn0, x.data_ptr(), N, h, w, b["y"][0], b["a"][0], None, True, s, conv_launch=launch)
                first_done = True
            res_bn = None
            if "down_conv" in blk and self.training:
                # shortcut conv + statistics only: its BatchNorm is applied inside the block's last BatchNorm pass, the
                # normalised shortcut is never stored (backward needs yd and the block mask, not it)
                _, scd, shd = self._conv_bn_fwd(ws, blk["down_conv"], blk["down_bn"], x.data_ptr(), N, h, w, b["yd"], None,
                                                None, False, s)
                idn, res_bn = b["yd"], (scd, shd)
            elif "down_conv" in blk:
                self._conv_bn_fwd(ws, blk["down_conv"], blk["down_bn"], x.data_ptr(), N, h, w, b["yd"], b["ad"], None,
                                  False, s)
                idn = b["ad"]
            else:
                idn = x
            cur, ch, cw = x, h, w
            for i, (conv, bn) in enumerate(zip(convs, bns)):
                last = i == len(convs) - 1
                if i == 0 and first_done:
                    d = conv.desc(N, ch, cw)
                    cur, ch, cw = b["a"][i], d.OH, d.OW
                    continue
                defer = False
                if last and self.training and self.block == "bottleneck" and bi + 1 < nblocks:
                    dn = self.blocks[bi + 1]["convs"][0].desc(N, ch, cw)
                    defer = bool(lib.icamd_bn_apply_conv1x1_fused_supported(ctypes.byref(dn)))
                if defer:
                    # conv3 + statistics + finalize now; the apply pass happens inside the next block's first convolution
                    d, sc3, sh3 = self._conv_bn_fwd(ws, conv, bn, cur.data_ptr(), N, ch, cw, b["y"][i], None, None, True, s)
                    pending = {"y": b["y"][i], "scale": sc3, "shift": sh3, "res": idn.data_ptr(),
                               "res_scale": res_bn[0] if res_bn else None, "res_shift": res_bn[1] if res_bn else None,
                               "out": b["a"][i], "mask": b["mask"], "name": bn.name}
                else:
                    d = self._conv_bn_fwd(ws, conv, bn, cur.data_ptr(), N, ch, cw, b["y"][i], b["a"][i],
                                          idn.data_ptr() if last else None, True, s,
                                          b["mask"].data_ptr() if (last and self.training) else None,
                                          res_bn if last else None)
                cur, ch, cw = b["a"][i], d.OH, d.OW
            x, h, w = cur, ch, cw
        hip.check(lib.icamd_avgpool_fwd(x.data_ptr(), ws["pooled"].data_ptr(), N, h * w, self.feat_dim, s), "avgpool")
        dfc = self.fc.desc(N, 1, 1)
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(dfc), ws["pooled"].data_ptr(), self._w(self.fc),
                                       ws["logits"].data_ptr(), self._pf(self.fc.b), None, None, s), "fc")
        return ws["logits"]

    def __call__(self, x_nchw):
        ws = self.pack(x_nchw.to(self.device, dtype=torch.float32).contiguous())
        logits = self.forward_packed(ws)
        return logits[:, : self.num_classes]

    # ------------------------------------------------------------------ backward
    def backward_packed(self, ws, accumulate=False):
        """Backward from ws['dlogits'] (bf16 [N, ncls_p]); fills the flat fp32 gradient arena.
        Gradients become final in reverse layer order; `grad_ready_hook(lo, hi)` is called as ranges complete.

        Weight gradients run on a second HIP stream (ICAMD_WGRAD_STREAM=0 keeps one stream): a layer's wgrad only needs
        that layer's output gradient and its saved input, nothing downstream needs its result before the optimizer, and it
        is MFMA/latency-bound while the BatchNorm-backward passes the main stream runs next are HBM-bound -- side by side
        they fill both.  The output-gradient buffers rotate through a small pool so the main stream can run ahead; an
        event per buffer keeps it from overwriting one a pending wgrad still reads."""
        if _FUSED_BNBWD:
            return self._backward_packed_fused(ws, accumulate)
        lib = self.lib
        main = torch.cuda.current_stream()
        s = main.cuda_stream
        N = ws["N"]
        acc = int(bool(accumulate))
        wsp, wsb = ws["wgrad_ws"].data_ptr(), ws["wgrad_ws_bytes"]
        bws, bwb = ws["bnb_ws"].data_ptr(), ws["bnb_ws_bytes"]
        bufs = self._grad_buffers(ws)
        D0, D1, T, DA = (b.data_ptr() for b in bufs[:4])
        ypool = [b.data_ptr() for b in bufs[4:]]
        hook = self.grad_ready_hook
        side = self._wgrad_stream() if (_WGRAD_STREAM and self.wgrad_side_stream) else None
        ws_side = side.cuda_stream if side is not None else s
        pending = [None] * len(ypool)     # per pool buffer: event of the last side-stream wgrad reading it
        state = {"next": 0, "last": None}

        def next_y():
            k = state["next"]
            state["next"] = (k + 1) % len(ypool)
            if pending[k] is not None:
                main.wait_event(pending[k])
                pending[k] = None
            return k

        def wgrad(conv, x_ptr, dy_ptr, n, ih, iw, ybuf=None):
            d = conv.desc(n, ih, iw)
            if side is not None:
                ready = torch.cuda.Event()
                ready.record(main)
                side.wait_event(ready)
            if conv is self.stem_conv:
                hip.check(lib.icamd_stem7x7s2_wgrad(x_ptr, dy_ptr, self._gf(conv.w), acc, wsp, wsb, n, ih, iw, conv.cout_p, ws_side),
                          conv.name + " wgrad")
            else:
                hip.check(lib.icamd_conv2d_wgrad(ctypes.byref(d), x_ptr, dy_ptr, self._gf(conv.w), acc, wsp, wsb, ws_side),
                          conv.name + " wgrad")
            if side is not None:
                done = torch.cuda.Event()
                done.record(side)
                state["last"] = done
                if ybuf is not None:
                    pending[ybuf] = done

        def join_side():
            if state["last"] is not None:
                main.wait_event(state["last"])
                state["last"] = None

        def side_events():
            """Events a gradient consumer on ANOTHER stream (the all-reduce) must wait for besides the main stream: the main
            stream itself keeps running ahead of the weight gradients."""
            return () if state["last"] is None else (state["last"],)

        def dgrad(conv, dy_ptr, dx_ptr, addend, n, ih, iw, addend_bits=None):
            d = conv.desc(n, ih, iw)
            hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy_ptr, self._wt(conv), dx_ptr, addend, addend_bits, s),
                      conv.name + " dgrad")

        def bn_bwd(bn, dout_ptr, act_ptr, y, dy_ptr, gout_ptr, relu, maskbits=None):
            st = self.stat_arena.data_ptr() + 4 * bn.stat_offset
            c = bn.c
            rows = y.numel() // c
            hip.check(lib.icamd_bn_bwd(dout_ptr, act_ptr, y.data_ptr(), st, st + 4 * c, st + 8 * c, st + 12 * c,
                                       self._gf(bn.weight), self._gf(bn.bias), dy_ptr, gout_ptr, maskbits, rows, c, int(relu),
                                       acc, bws, bwb, s), bn.name + " bwd")

        # classifier
        dl = ws["dlogits"].data_ptr()
        wgrad(self.fc, ws["pooled"].data_ptr(), dl, N, 1, 1)
        hip.check(lib.icamd_colsum(dl, N, self.ncls_p, self.ncls_p, self._gf(self.fc.b), acc, s), "fc bias grad")
        dgrad(self.fc, dl, ws["dpooled"].data_ptr(), None, N, 1, 1)
        if hook:
            hook(self.fc.w.offset, self.n_params, side_events())
        fh, fw = ws["final_hw"]
        dout, other = D0, D1
        hip.check(lib.icamd_avgpool_bwd(ws["dpooled"].data_ptr(), dout, N, fh * fw, self.feat_dim, s), "avgpool bwd")

        # Residual data gradients that ALSO do pass 1 of the previous block's last BatchNorm backward (icamd_conv2d_dgrad_bnred):
        # `fused_rows` > 0 says `dout` already holds g = masked output gradient of the block about to be processed and
        # ws["bnb_part"] its partial sums (sum g, sum g*y), so that block starts with the apply pass only
        fused_rows = 0
        for bi in range(len(self.blocks) - 1, -1, -1):
            blk, b = self.blocks[bi], ws["blocks"][bi]
            convs, bns = blk["convs"], blk["bns"]
            h, w = b["in_hw"]
            xin = b["in"]
            nconv = len(convs)
            # spatial sizes seen by each conv's input
            hw_in = [(h, w)]
            for conv in convs[:-1]:
                d = conv.desc(N, *hw_in[-1])
                hw_in.append((d.OH, d.OW))
            # last BN (+ residual + ReLU): g = dout * [block output > 0] via the 1-bit mask the forward stored; g itself is
            # never written: the shortcut consumers below re-apply the same bits to `dout`
            mask = b["mask"].data_ptr()
            yk = next_y()
            y2 = None
            fused_conv3 = False
            shortcut_done = False
            # may this block's conv1 data gradient hand the PREVIOUS block g = its masked output gradient plus the (sum g, sum g*y)
            # rows of its last BatchNorm (icamd_conv2d_dgrad_bnred)?  Identity blocks always take them; a projection block only
            # through the fused conv3 + bn3 backward (its other path, icamd_bn_bwd_dual, wants the unmasked gradient and no sums)
            prev_takes_g = False
            if bi > 0 and self.block == "bottleneck":
                pblk = self.blocks[bi - 1]
                if "down_conv" not in pblk:
                    prev_takes_g = True
                else:
                    dp3 = pblk["convs"][-1].desc(N, h, w)
                    prev_takes_g = bool(lib.icamd_conv1x1_bn_bwd_fused_supported(ctypes.byref(dp3)))
            d3 = convs[-1].desc(N, *hw_in[-1])
            fuse3 = (self.block == "bottleneck" and fused_rows > 0
                     and bool(lib.icamd_conv1x1_bn_bwd_fused_supported(ctypes.byref(d3))))

            def fused_conv_bn(conv, bn, dsc, partials, nrows, y_t, x_ptr, dx_ptr, bn_ws, bn_ws_bytes):
                """BatchNorm-backward apply + data gradient + weight gradient of `conv` -> `bn` in one pass over g (= dout, already
                masked) and y (round 5, icamd_conv1x1_bn_bwd_fused); main stream, its own slab workspace (the side stream's weight
                gradients own ws["wgrad_ws"])."""
                st_ = self.stat_arena.data_ptr() + 4 * bn.stat_offset
                c_ = bn.c
                need = lib.icamd_conv1x1_bn_bwd_fused_workspace_bytes(ctypes.byref(dsc))
                fws = ws.get("fused_ws")
                if fws is None or fws.numel() < need:
                    fws = ws["fused_ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
                hip.check(lib.icamd_conv1x1_bn_bwd_fused(ctypes.byref(dsc), partials, nrows, dout, y_t.data_ptr(), st_, st_ + 4 * c_,
                                                         st_ + 8 * c_, self._gf(bn.weight), self._gf(bn.bias), x_ptr, self._wt(conv),
                                                         dx_ptr, self._gf(conv.w), acc, bn_ws, bn_ws_bytes, fws.data_ptr(),
                                                         fws.numel(), s), bn.name + " + " + conv.name + " bwd (fused)")

            if "down_conv" in blk and fuse3:
                # projection block whose successor left g and the (sum g, sum g*y3) rows: conv3 + bn3 fused; the shortcut's BatchNorm
                # takes the same g -- its convolution + BatchNorm fused as well where it is a 1x1 / stride-1 layer of a routed
                # shape (layer1.0), else its BatchNorm backward alone (no mask: g is masked already)
                fused_conv_bn(convs[-1], bns[-1], d3, ws["bnb_part"].data_ptr(), fused_rows, b["y"][-1],
                              b["a"][nconv - 2].data_ptr(), DA, ws["bna_ws"].data_ptr(), ws["bna_ws_bytes"])
                fused_conv3 = True
                dc = blk["down_conv"]
                ddc = dc.desc(N, h, w)
                if dc.stride == 1 and lib.icamd_conv1x1_bn_bwd_fused_supported(ctypes.byref(ddc)):
                    fused_conv_bn(dc, blk["down_bn"], ddc, None, 0, b["yd"], xin.data_ptr(), T, bws, bwb)
                    shortcut_done = True
                else:
                    y2 = next_y()
                    bn_bwd(blk["down_bn"], dout, None, b["yd"], ypool[y2], None, False)
            elif "down_conv" in blk and _DUAL_BNBWD:
                # the block's last BatchNorm and its shortcut's BatchNorm take the same masked gradient: one reduce and one
                # apply pass for both (dout and the mask bits are read twice instead of four times)
                y2 = next_y()
                bnA, bnB = bns[-1], blk["down_bn"]
                stA = self.stat_arena.data_ptr() + 4 * bnA.stat_offset
                stB = self.stat_arena.data_ptr() + 4 * bnB.stat_offset
                c = bnA.c
                hip.check(lib.icamd_bn_bwd_dual(dout, mask, b["y"][-1].data_ptr(), stA, stA + 4 * c, stA + 8 * c,
                                                self._gf(bnA.weight), self._gf(bnA.bias), ypool[yk], b["yd"].data_ptr(), stB,
                                                stB + 4 * c, stB + 8 * c, self._gf(bnB.weight), self._gf(bnB.bias), ypool[y2],
                                                b["y"][-1].numel() // c, c, acc, bws, ws["bnb_ws2"].data_ptr(), bwb, s),
                          bnA.name + " + shortcut bwd")
            elif fused_rows:
                bnl = bns[-1]
                stl = self.stat_arena.data_ptr() + 4 * bnl.stat_offset
                cl = bnl.c
                if fuse3:
                    # conv3 + bn3 backward in one pass over g and y3: BatchNorm finalize from the partial sums, dy3 only in LDS,
                    # d(a2) -> DA and the filter gradient out of the same launch
                    fused_conv_bn(convs[-1], bnl, d3, ws["bnb_part"].data_ptr(), fused_rows, b["y"][-1],
                                  b["a"][nconv - 2].data_ptr(), DA, ws["bna_ws"].data_ptr(), ws["bna_ws_bytes"])
                    fused_conv3 = True
                else:
                    hip.check(lib.icamd_bn_bwd_from_gy_partials(ws["bnb_part"].data_ptr(), fused_rows, dout, b["y"][-1].data_ptr(),
                                                                stl, stl + 4 * cl, stl + 8 * cl, self._gf(bnl.weight),
                                                                self._gf(bnl.bias), ypool[yk], b["y"][-1].numel() // cl, cl, acc,
                                                                ws["bna_ws"].data_ptr(), ws["bna_ws_bytes"], s),
                              bnl.name + " bwd (apply, sums from the data gradient)")
            else:
                bn_bwd(bns[-1], dout, None, b["y"][-1], ypool[yk], None, True, mask)
            fused_rows = 0
            for i in range(nconv - 1, 0, -1):
                x_i = b["a"][i - 1]
                if fused_conv3 and i == nconv - 1:
                    # conv3's data and weight gradients came out of the fused launch (DA, grad arena)
                    yk = next_y()
                    bn_bwd(bns[i - 1], DA, None, b["y"][i - 1], ypool[yk], None, True)
                    continue
                # wgrad first: measured, it overlaps best with the data-gradient kernel of the same layer (issued after it,
                # i.e. beside the next BatchNorm backward whose 1024 workgroups fill every wave slot, the gain disappears)
                wgrad(convs[i], x_i.data_ptr(), ypool[yk], N, *hw_in[i], ybuf=yk)
                dgrad(convs[i], ypool[yk], DA, None, N, *hw_in[i])
                # BN + ReLU with no residual in front of the ReLU: mask recomputed from y
                yk = next_y()
                bn_bwd(bns[i - 1], DA, None, b["y"][i - 1], ypool[yk], None, True)
            wgrad(convs[0], xin.data_ptr(), ypool[yk], N, h, w, ybuf=yk)
            if shortcut_done:
                # the shortcut's filter gradient is done and T holds its (full-size) data gradient
                dgrad(convs[0], ypool[yk], other, T, N, h, w)
            elif "down_conv" in blk:
                if y2 is None:
                    y2 = next_y()
                    bn_bwd(blk["down_bn"], dout, None, b["yd"], ypool[y2], None, True, mask)   # "relu" = the block's mask bits
                dc = blk["down_conv"]
                wgrad(dc, xin.data_ptr(), ypool[y2], N, h, w, ybuf=y2)
                if _SUB2_SHORTCUT and dc.k == 1 and dc.stride == 2 and dc.pad == 0:
                    # a 1x1 stride-2 shortcut sends gradient to the even pixels only: compute it on the [OH][OW] grid (a
                    # plain pointwise data gradient) and let the main branch's data gradient add it there
                    # (icamd_conv2d_dgrad_sub2); the 3/4 zeros of the full-size tensor are never written or read
                    dd = dc.desc(N, h, w)
                    key = ("sub2", N, dd.OH, dd.OW)
                    d1 = dc.descs.get(key)
                    if d1 is None:
                        d1 = dc.descs[key] = hip.conv_desc(N, dd.OH, dd.OW, dc.cin_p, dc.cout_p, 1, 1, 1, 0)
                    hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d1), ypool[y2], self._wt(dc), T, None, None, s),
                              dc.name + " dgrad (even grid)")
                    d0 = convs[0].desc(N, h, w)
                    if (prev_takes_g and self.block == "bottleneck"
                            and lib.icamd_conv2d_dgrad_bnred_supported(ctypes.byref(d0))):
                        pb = ws["blocks"][bi - 1]   # (see the identity-shortcut case below)
                        hip.check(lib.icamd_conv2d_dgrad_bnred(ctypes.byref(d0), ypool[yk], self._wt(convs[0]), other, T, None, 1,
                                                               pb["y"][-1].data_ptr(), pb["mask"].data_ptr(),
                                                               ws["bnb_part"].data_ptr(), s),
                                  convs[0].name + " dgrad + shortcut + bn reduce")
                        fused_rows = lib.icamd_conv2d_dgrad_stats_rows(ctypes.byref(d0))
                    else:
                        hip.check(lib.icamd_conv2d_dgrad_sub2(ctypes.byref(d0), ypool[yk], self._wt(convs[0]), other, T, s),
                                  convs[0].name + " dgrad + shortcut")
                else:
                    dgrad(dc, ypool[y2], T, None, N, h, w)
                    dgrad(convs[0], ypool[yk], other, T, N, h, w)
            else:
                d0 = convs[0].desc(N, h, w)
                if (prev_takes_g and self.block == "bottleneck"
                        and lib.icamd_conv2d_dgrad_bnred_supported(ctypes.byref(d0))):
                    # `other` becomes g of the previous block (its ReLU mask applied, which every consumer of d(block output)
                    # applies anyway) and the sums its last BatchNorm's backward needs come out of the same launch
                    pb = ws["blocks"][bi - 1]
                    hip.check(lib.icamd_conv2d_dgrad_bnred(ctypes.byref(d0), ypool[yk], self._wt(convs[0]), other, dout, mask, 0,
                                                           pb["y"][-1].data_ptr(), pb["mask"].data_ptr(),
                                                           ws["bnb_part"].data_ptr(), s), convs[0].name + " dgrad + bn reduce")
                    fused_rows = lib.icamd_conv2d_dgrad_stats_rows(ctypes.byref(d0))
                else:
                    dgrad(convs[0], ypool[yk], other, dout, N, h, w, mask)
            if hook:
                hook(convs[0].w.offset, None, side_events())
            dout, other = other, dout

        # stem: maxpool -> BN+ReLU -> conv (no data gradient for the image)
        d0 = self.stem_conv.desc(N, ws["H"], ws["Ws"])
        yk = next_y()
        if _FUSED_POOL_BWD:
            # max-pool backward folded into both BatchNorm-backward passes: the 112x112 gradient is never materialised
            bn0 = self.stem_bn
            st = self.stat_arena.data_ptr() + 4 * bn0.stat_offset
            c = bn0.c
            hip.check(lib.icamd_bn_bwd_maxpool3x3s2(dout, ws["p0_idx"].data_ptr(), ws["y0"].data_ptr(), st, st + 4 * c,
                                                    st + 8 * c, st + 12 * c, self._gf(bn0.weight), self._gf(bn0.bias),
                                                    ypool[yk], N, d0.OH, d0.OW, c, acc, bws, bwb, s), "stem maxpool + bn bwd")
        else:
            hip.check(lib.icamd_maxpool3x3s2_bwd(dout, ws["p0_idx"].data_ptr(), DA, N, d0.OH, d0.OW, 64, s), "maxpool bwd")
            bn_bwd(self.stem_bn, DA, None, ws["y0"], ypool[yk], None, True)
        wgrad(self.stem_conv, ws["x8"].data_ptr(), ypool[yk], N, ws["H"], ws["Ws"], ybuf=yk)
        for k in range(len(pending)):     # every buffer is free again when the next backward starts
            pending[k] = None
        if hook:
            hook(0, None, side_events())
        join_side()

    def _wgrad_stream(self):
        if getattr(self, "_wg_stream", None) is None:
            self._wg_stream = torch.cuda.Stream(device=self.device)
        return self._wg_stream

    def _backward_packed_fused(self, ws, accumulate=False):
        """Variant that fuses each BatchNorm backward's mask + reduction pass into the epilogue of the data-gradient
        kernel that produces its output gradient (icamd_conv2d_dgrad_bnbwd + icamd_bn_bwd_from_partials).  Fewer HBM
        bytes, but the un-pipelined epilogue makes it latency-bound on MI355X today (profiles/ r01 notes): opt-in with
        ICAMD_FUSED_BNBWD=1."""
        lib = self.lib
        s = hip.stream_ptr()
        N = ws["N"]
        acc = int(bool(accumulate))
        wsp, wsb = ws["wgrad_ws"].data_ptr(), ws["wgrad_ws_bytes"]
        bws, bwb = ws["bnb_ws"].data_ptr(), ws["bnb_ws_bytes"]
        D0, D1, G, T, Y, Y2, DA = (b.data_ptr() for b in self._grad_buffers(ws)[:7])
        hook = self.grad_ready_hook

        def wgrad(conv, x_ptr, dy_ptr, n, ih, iw):
            d = conv.desc(n, ih, iw)
            if conv is self.stem_conv:
                hip.check(lib.icamd_stem7x7s2_wgrad(x_ptr, dy_ptr, self._gf(conv.w), acc, wsp, wsb, n, ih, iw, conv.cout_p, s),
                          conv.name + " wgrad")
                return
            hip.check(lib.icamd_conv2d_wgrad(ctypes.byref(d), x_ptr, dy_ptr, self._gf(conv.w), acc, wsp, wsb, s),
                      conv.name + " wgrad")

        def dgrad(conv, dy_ptr, dx_ptr, addend, n, ih, iw):
            d = conv.desc(n, ih, iw)
            hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy_ptr, self._wt(conv), dx_ptr, addend, None, s),
                      conv.name + " dgrad")

        def bn_bwd(bn, dout_ptr, act_ptr, y, dy_ptr, gout_ptr, relu):
            st = self.stat_arena.data_ptr() + 4 * bn.stat_offset
            c = bn.c
            rows = y.numel() // c
            hip.check(lib.icamd_bn_bwd(dout_ptr, act_ptr, y.data_ptr(), st, st + 4 * c, st + 8 * c, st + 12 * c,
                                       self._gf(bn.weight), self._gf(bn.bias), dy_ptr, gout_ptr, None, rows, c, int(relu), acc,
                                       bws, bwb, s), bn.name + " bwd")

        P, pw_bytes = ws["bnb_part"].data_ptr(), ws["bna_ws_bytes"]
        aws = ws["bna_ws"].data_ptr()

        def dgrad_bnbwd(conv, dy_ptr, g_ptr, addend, n, ih, iw, bn, y, mask_src, relu):
            """data gradient whose output is the output-gradient of `bn`(+ReLU): fused mask + pass-1 reductions."""
            d = conv.desc(n, ih, iw)
            st = self.stat_arena.data_ptr() + 4 * bn.stat_offset
            c = bn.c
            f = hip.BnBwdFuse(y.data_ptr(), mask_src, st, st + 4 * c, st + 8 * c, st + 12 * c, P, int(relu))
            hip.check(lib.icamd_conv2d_dgrad_bnbwd(ctypes.byref(d), dy_ptr, self._wt(conv), g_ptr, addend, ctypes.byref(f), s),
                      conv.name + " dgrad+bnbwd")
            return lib.icamd_conv2d_dgrad_stats_rows(ctypes.byref(d))

        def bn_bwd_from_partials(bn, nrows, g_ptr, y, dy_ptr):
            st = self.stat_arena.data_ptr() + 4 * bn.stat_offset
            c = bn.c
            hip.check(lib.icamd_bn_bwd_from_partials(P, nrows, g_ptr, y.data_ptr(), st, st + 4 * c, st + 8 * c,
                                                     self._gf(bn.weight), self._gf(bn.bias), dy_ptr, y.numel() // c, c, acc,
                                                     aws, pw_bytes, s), bn.name + " bwd(apply)")

        # classifier
        dl = ws["dlogits"].data_ptr()
        wgrad(self.fc, ws["pooled"].data_ptr(), dl, N, 1, 1)
        hip.check(lib.icamd_colsum(dl, N, self.ncls_p, self.ncls_p, self._gf(self.fc.b), acc, s), "fc bias grad")
        dgrad(self.fc, dl, ws["dpooled"].data_ptr(), None, N, 1, 1)
        if hook:
            hook(self.fc.w.offset, self.n_params)
        fh, fw = ws["final_hw"]
        dout, other = D0, D1
        hip.check(lib.icamd_avgpool_bwd(ws["dpooled"].data_ptr(), dout, N, fh * fw, self.feat_dim, s), "avgpool bwd")

        nblk = len(self.blocks)
        pending_rows = 0   # partial rows left in P by the previous block's fused data gradient (for this block's last BN)
        for bi in range(nblk - 1, -1, -1):
            blk, b = self.blocks[bi], ws["blocks"][bi]
            convs, bns = blk["convs"], blk["bns"]
            h, w = b["in_hw"]
            xin = b["in"]
            nconv = len(convs)
            hw_in = [(h, w)]
            for conv in convs[:-1]:
                d = conv.desc(N, *hw_in[-1])
                hw_in.append((d.OH, d.OW))
            if bi == nblk - 1:
                # last block: its output gradient comes from the average pool: two-pass BN backward with the mask
                bn_bwd(bns[-1], dout, b["a"][-1].data_ptr(), b["y"][-1], Y, G, True)
                g_ptr = G
            else:
                # `dout` already holds g = masked output gradient and P its pass-1 sums (fused in the producer)
                bn_bwd_from_partials(bns[-1], pending_rows, dout, b["y"][-1], Y)
                g_ptr = dout
            for i in range(nconv - 1, 0, -1):
                x_i = b["a"][i - 1]
                wgrad(convs[i], x_i.data_ptr(), Y, N, *hw_in[i])
                nrows = dgrad_bnbwd(convs[i], Y, DA, None, N, *hw_in[i], bns[i - 1], b["y"][i - 1], None, True)
                bn_bwd_from_partials(bns[i - 1], nrows, DA, b["y"][i - 1], Y)
            wgrad(convs[0], xin.data_ptr(), Y, N, h, w)
            if "down_conv" in blk:
                bn_bwd(blk["down_bn"], g_ptr, None, b["yd"], Y2, None, False)
                wgrad(blk["down_conv"], xin.data_ptr(), Y2, N, h, w)
                dgrad(blk["down_conv"], Y2, T, None, N, h, w)
                addend = T
            else:
                addend = g_ptr
            if bi > 0:
                pb, pblk = ws["blocks"][bi - 1], self.blocks[bi - 1]
                pending_rows = dgrad_bnbwd(convs[0], Y, other, addend, N, h, w, pblk["bns"][-1], pb["y"][-1],
                                           xin.data_ptr(), True)
            else:
                dgrad(convs[0], Y, other, addend, N, h, w)
            if hook:
                hook(convs[0].w.offset, None)
            dout, other = other, dout

        # stem: maxpool -> BN+ReLU -> conv (no data gradient for the image)
        d0 = self.stem_conv.desc(N, ws["H"], ws["Ws"])
        hip.check(lib.icamd_maxpool3x3s2_bwd(dout, ws["p0_idx"].data_ptr(), DA, N, d0.OH, d0.OW, 64, s), "maxpool bwd")
        bn_bwd(self.stem_bn, DA, None, ws["y0"], Y, None, True)
        wgrad(self.stem_conv, ws["x8"].data_ptr(), Y, N, ws["H"], ws["Ws"])
        if hook:
            hook(0, None)
