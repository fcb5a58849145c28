"""ViT-B/16 (timm `vit_base_patch16_224`) on the gfx950 kernels: hand-written forward and backward.

The reference builds the model with timm.create_model(args.model) (/root/reference/train.py:194); BASELINE.json's
configs[3] is ViT-B/16 bf16 at 224x224.  Architecture restated from timm (absent here, see oracle/vit_ref.py):
16x16/16 patch-embedding conv + bias, class token, learned position embedding, 12 pre-LayerNorm blocks
(LayerNorm eps 1e-6, 12-head attention with fused QKV projection, exact-GELU MLP x4), final LayerNorm, class-token
pooling, linear head.  Parameter names follow timm (`cls_token`, `pos_embed`, `patch_embed.proj.*`,
`blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*`, `norm.*`, `head.*`).

Every Linear is the 1x1 case of the implicit-GEMM convolution kernels on a [B*T, 1, 1, C] "image" (bias and the
residual add fused in the epilogue); LayerNorm / GELU / attention are the token kernels of include/icamd.h.
Same flat-arena design as nets.ResNet (fp32 parameters and gradients, bf16 shadow weights and their transposes).
"""
import ctypes
import os
import math
from collections import OrderedDict

import torch

from . import hip
from .checkpoint import PicklableModel

LN_EPS = 1e-6

CONFIGS = {
    # name: (patch, dim, depth, heads, mlp_ratio)
    "vit_base_patch16_224": (16, 768, 12, 12, 4),
    "vit_small_patch16_224": (16, 384, 12, 6, 4),
    "vit_tiny_test": (16, 128, 2, 2, 4),   # small configuration for parity tests
}


def _align(n, a):
    return (n + a - 1) // a * a


class _P:
    __slots__ = ("name", "offset", "numel", "torch_shape", "kind", "padded_shape")

    def __init__(self, name, offset, numel, torch_shape, kind, padded_shape):
        self.name, self.offset, self.numel = name, offset, numel
        self.torch_shape, self.kind, self.padded_shape = torch_shape, kind, padded_shape


class _Lin:
    """Linear layer = 1x1 convolution record (weight [out_p][in] in the arena)."""

    def __init__(self, name, cin, cout, cout_p=None):
        self.name, self.cin, self.cout = name, cin, cout
        self.cout_p = cout_p or cout
        self.w = self.b = None
        self.wt_offset = None
        self.descs = {}

    def desc(self, rows):
        d = self.descs.get(rows)
        if d is None:
            d = hip.conv_desc(rows, 1, 1, self.cin, self.cout_p, 1, 1, 1, 0)
            self.descs[rows] = d
        return d


class VisionTransformer(PicklableModel):
    def __init__(self, arch="vit_base_patch16_224", num_classes=1000, device="cuda", img_size=224, seed=None):
        hip.require_gpu()
        self.lib = hip.load()
        self.arch, self.num_classes = arch, num_classes
        self.device = torch.device(device)
        self.training = True
        self.patch, self.dim, self.depth, self.heads, mlp_ratio = CONFIGS[arch]
        if self.dim // self.heads != 64:
            raise ValueError("the attention kernel is built for a head dimension of 64")
        self.hidden = self.dim * mlp_ratio
        self.img_size = img_size
        self.grid = img_size // self.patch
        self.T = self.grid * self.grid + 1
        self.ncls_p = _align(num_classes, 64)
        self.num_batches_tracked = 0
        self.grad_ready_hook = None
        self._ws = {}
        self._build()
        self.init_weights(seed)

    # ------------------------------------------------------------------ structure / arenas
    def _build(self):
        dev = self.device
        D = self.dim
        self.params = OrderedDict()
        off = 0

        def add(name, torch_shape, kind, padded_shape):
            nonlocal off
            numel = 1
            for s in padded_shape:
                numel *= s
            p = _P(name, off, numel, tuple(torch_shape), kind, tuple(padded_shape))
            self.params[name] = p
            off = _align(off + numel, 64)
            return p

        self.lins = []

        def lin(name, cin, cout, cout_p=None):
            l = _Lin(name, cin, cout, cout_p)
            l.w = add(name + ".weight", (cout, cin), "lin", (l.cout_p, cin))
            l.b = add(name + ".bias", (cout,), "vec", (l.cout_p,))
            self.lins.append(l)
            return l

        self.p_cls = add("cls_token", (1, 1, D), "vec", (D,))
        self.p_pos = add("pos_embed", (1, self.T, D), "vec", (self.T * D,))
        self.pe_w = add("patch_embed.proj.weight", (D, 3, self.patch, self.patch), "conv", (D, self.patch, self.patch, 8))
        self.pe_b = add("patch_embed.proj.bias", (D,), "vec", (D,))
        self.blocks = []
        for i in range(self.depth):
            n = f"blocks.{i}"
            blk = {"name": n}
            blk["n1w"] = add(f"{n}.norm1.weight", (D,), "vec", (D,))
            blk["n1b"] = add(f"{n}.norm1.bias", (D,), "vec", (D,))
            blk["qkv"] = lin(f"{n}.attn.qkv", D, 3 * D)
            blk["proj"] = lin(f"{n}.attn.proj", D, D)
            blk["n2w"] = add(f"{n}.norm2.weight", (D,), "vec", (D,))
            blk["n2b"] = add(f"{n}.norm2.bias", (D,), "vec", (D,))
            blk["fc1"] = lin(f"{n}.mlp.fc1", D, self.hidden)
            blk["fc2"] = lin(f"{n}.mlp.fc2", self.hidden, D)
            self.blocks.append(blk)
        self.p_nw = add("norm.weight", (D,), "vec", (D,))
        self.p_nb = add("norm.bias", (D,), "vec", (D,))
        self.head = lin("head", D, self.num_classes, self.ncls_p)
        self.n_params = off
        self.param_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.buffer_arena = torch.zeros(64, dtype=torch.float32, device=dev)   # no buffers; kept for the EMA/DDP protocol
        toff, descs, tjobs = 0, [], []
        for l in self.lins:
            l.wt_offset = toff
            descs.append([l.w.offset, toff, l.cout_p, 1, l.cin, 0, 0, 0])
            for co0 in range(0, l.cout_p, 64):
                for ci0 in range(0, l.cin, 64):
                    tjobs.append([len(descs) - 1, 0, co0, ci0])
            toff = _align(toff + l.w.numel, 128)
        self.shadow_t = torch.zeros(toff, dtype=torch.bfloat16, device=dev)
        self._tr_descs = torch.tensor(descs, dtype=torch.int64, device=dev)
        self._tr_tjobs = torch.tensor(tjobs, dtype=torch.int32, device=dev)
        self._tr_ntjobs = len(tjobs)

    def _ctor_kwargs(self):
        return {"arch": self.arch, "num_classes": self.num_classes, "img_size": self.img_size}

    def init_weights(self, seed=None):
        """timm's default ViT init: trunc_normal(std .02) weights / pos_embed, zero biases, cls_token std 1e-6, LayerNorm
        1 / 0; the patch-embedding conv keeps torch's Conv2d default (Kaiming-uniform, a = sqrt(5))."""
        g = torch.Generator()
        g.manual_seed(seed if seed is not None else torch.initial_seed() % (2 ** 63))
        sd = OrderedDict()

        def tn(shape, std):
            return torch.nn.init.trunc_normal_(torch.empty(shape), std=std, generator=g)

        for name, p in self.params.items():
            if name == "cls_token":
                sd[name] = torch.randn(p.torch_shape, generator=g) * 1e-6
            elif name == "pos_embed":
                sd[name] = tn(p.torch_shape, 0.02)
            elif name == "patch_embed.proj.weight":
                fan_in = 3 * self.patch * self.patch
                bound = 1.0 / math.sqrt(fan_in)
                sd[name] = (torch.rand(p.torch_shape, generator=g) * 2 - 1) * bound
            elif name == "patch_embed.proj.bias":
                bound = 1.0 / math.sqrt(3 * self.patch * self.patch)
                sd[name] = (torch.rand(p.torch_shape, generator=g) * 2 - 1) * bound
            elif p.kind == "lin":
                sd[name] = tn(p.torch_shape, 0.02)
            elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
                sd[name] = torch.ones(p.torch_shape)
            else:
                sd[name] = torch.zeros(p.torch_shape)
        self.load_state_dict(sd)

    def _to_arena(self, p, t):
        t = t.detach().to(torch.float32).cpu()
        full = torch.zeros(p.padded_shape)
        if p.kind == "conv":
            cout, cin = t.shape[0], t.shape[1]
            full[:cout, :, :, :cin] = t.permute(0, 2, 3, 1)
        elif p.kind == "lin":
            full[: t.shape[0], :] = t
        else:
            full.view(-1)[: t.numel()] = t.flatten()
        return full.flatten()

    def _from_arena(self, p, flat):
        t = flat.reshape(p.padded_shape)
        if p.kind == "conv":
            return t[: p.torch_shape[0], :, :, : p.torch_shape[1]].permute(0, 3, 1, 2).contiguous()
        if p.kind == "lin":
            return t[: p.torch_shape[0], :].clone()
        n = 1
        for s in p.torch_shape:
            n *= s
        return t.flatten()[:n].reshape(p.torch_shape).clone()

    def load_state_dict(self, sd, strict=True):
        host = self.param_arena.cpu()
        missing = [n for n in self.params if n not in sd]
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing[:5]}")
        for name, p in self.params.items():
            if name in sd:
                host[p.offset:p.offset + p.numel] = self._to_arena(p, sd[name])
        self.param_arena.copy_(host)
        self.refresh_shadow()
        return missing

    def state_dict(self):
        host = self.param_arena.cpu()
        return OrderedDict((n, self._from_arena(p, host[p.offset:p.offset + p.numel])) for n, p in self.params.items())

    def named_parameters(self):
        for name, p in self.params.items():
            yield name, self.param_arena[p.offset:p.offset + p.numel]

    def parameters(self):
        for _, v in self.named_parameters():
            yield v

    def grad_of(self, name):
        p = self.params[name]
        return self._from_arena(p, self.grad_arena[p.offset:p.offset + p.numel].cpu())

    def refresh_shadow(self):
        hip.check(self.lib.icamd_f32_to_bf16(self.param_arena.data_ptr(), self.shadow.data_ptr(), self.n_params,
                                             hip.stream_ptr()), "f32_to_bf16")
        self.refresh_transposed()

    def refresh_transposed(self):
        hip.check(self.lib.icamd_filter_transpose_tiled(self.shadow.data_ptr(), self.shadow_t.data_ptr(),
                                                        self._tr_descs.data_ptr(), self._tr_tjobs.data_ptr(), self._tr_ntjobs,
                                                        hip.stream_ptr()), "filter_transpose_tiled")

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B):
        ws = self._ws.get(B)
        if ws is not None:
            return ws
        dev, lib = self.device, self.lib
        D, T, Hd = self.dim, self.T, self.hidden
        M = B * T

        def act(r, c):
            return torch.empty(r, c, dtype=torch.bfloat16, device=dev)

        ws = {"B": B, "M": M}
        ws["x8"] = torch.empty(B, self.img_size, self.img_size, 8, dtype=torch.bfloat16, device=dev)
        ws["patches"] = act(B * (T - 1), D)
        ws["x0"] = act(M, D)
        blocks = []
        for _ in range(self.depth):
            blocks.append({"h": act(M, D), "qkv": act(M, 3 * D), "ao": act(M, D), "x1": act(M, D), "h2": act(M, D),
                           "z": act(M, Hd), "a": act(M, Hd), "x2": act(M, D),
                           "lse": torch.empty(B * self.heads * T, dtype=torch.float32, device=dev),
                           "st1": torch.empty(2 * M, dtype=torch.float32, device=dev),
                           "st2": torch.empty(2 * M, dtype=torch.float32, device=dev)})
        ws["blocks"] = blocks
        ws["cls_rows"] = act(B, D)
        ws["pooled"] = act(B, D)
        ws["stf"] = torch.empty(2 * B, dtype=torch.float32, device=dev)
        ws["logits"] = torch.zeros(B, self.ncls_p, dtype=torch.bfloat16, device=dev)
        ws["dlogits"] = torch.zeros(B, self.ncls_p, dtype=torch.bfloat16, device=dev)
        ws["loss_rows"] = torch.empty(B, dtype=torch.float32, device=dev)
        ws["pred"] = torch.empty(B, dtype=torch.int32, device=dev)
        # backward scratch
        ws["g768"] = [act(M, D) for _ in range(3)]
        ws["g3072"] = act(M, Hd)
        ws["g3072b"] = act(M, Hd)
        ws["g2304"] = act(M, 3 * D)
        ws["dpooled"] = act(B, D)
        ws["dcls"] = act(B, D)
        ws["delta"] = torch.empty(B * self.heads * T, dtype=torch.float32, device=dev)
        wg = 0
        for l in self.lins:
            wg = max(wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(l.desc(B if l is self.head else M))))
        dpe = self._pe_desc(B)
        wg = max(wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(dpe)))
        ws["wg_ws"] = torch.empty(wg, dtype=torch.uint8, device=dev)
        ws["wg_bytes"] = wg
        ws["ln_bytes"] = lib.icamd_layernorm_bwd_workspace_bytes(M, D)
        ws["ln_ws"] = torch.zeros(ws["ln_bytes"], dtype=torch.uint8, device=dev)
        ws["cs_bytes"] = lib.icamd_colsum_rows_workspace_bytes(M, 3 * D if 3 * D > Hd else Hd)
        ws["cs_ws"] = torch.zeros(ws["cs_bytes"], dtype=torch.uint8, device=dev)
        self._ws[B] = ws
        return ws

    def _pe_desc(self, B):
        key = ("pe", B)
        d = self._ws.get(key)
        if d is None:
            d = hip.conv_desc(B, self.img_size, self.img_size, 8, self.dim, self.patch, self.patch, self.patch, 0)
            self._ws[key] = d
        return d

    # ------------------------------------------------------------------ helpers
    def _pf(self, p):
        return self.param_arena.data_ptr() + 4 * p.offset

    def _gf(self, p):
        return self.grad_arena.data_ptr() + 4 * p.offset

    def _w(self, l):
        return self.shadow.data_ptr() + 2 * l.w.offset

    def _wt(self, l):
        return self.shadow_t.data_ptr() + 2 * l.wt_offset

    def pack(self, x_nchw, mix=None):
        B, C, H, W = x_nchw.shape
        assert H == self.img_size and W == self.img_size, "ViT position embedding is built for a fixed input size"
        ws = self._workspace(B)
        mode, lam, box = (0, 1.0, (0, 0, 0, 0)) if mix is None else mix
        hip.check(self.lib.icamd_pack_input(x_nchw.data_ptr(), ws["x8"].data_ptr(), B, C, H, W, mode, float(lam), int(box[0]),
                                            int(box[1]), int(box[2]), int(box[3]), hip.stream_ptr()), "pack")
        return ws

    def _linear(self, l, x, y, rows, addend, s):
        hip.check(self.lib.icamd_conv2d_fwd(ctypes.byref(l.desc(rows)), x.data_ptr(), self._w(l), y.data_ptr(), self._pf(l.b),
                                            None if addend is None else addend.data_ptr(), None, s), l.name)

    # ------------------------------------------------------------------ forward
    def forward_packed(self, ws, logits_only=False):
        """logits_only: a forward whose activations no backward will read (the reference's second, accuracy-only forward under
        mixup): tensors kept only for the backward pass (the pre-GELU Mlp activations) are not written."""
        lib, s = self.lib, hip.stream_ptr()
        B, M, D, T = ws["B"], ws["M"], self.dim, self.T
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(self._pe_desc(B)), ws["x8"].data_ptr(),
                                       self.shadow.data_ptr() + 2 * self.pe_w.offset, ws["patches"].data_ptr(),
                                       self._pf(self.pe_b), None, None, s), "patch_embed")
        hip.check(lib.icamd_vit_tokens_fwd(ws["patches"].data_ptr(), self._pf(self.p_cls), self._pf(self.p_pos),
                                           ws["x0"].data_ptr(), B, T, D, s), "tokens")
        x = ws["x0"]
        scale = 64 ** -0.5
        for blk, b in zip(self.blocks, ws["blocks"]):
            b["x"] = x
            hip.check(lib.icamd_layernorm_fwd(x.data_ptr(), self._pf(blk["n1w"]), self._pf(blk["n1b"]), b["h"].data_ptr(),
                                              b["st1"].data_ptr(), b["st1"].data_ptr() + 4 * M, M, D, LN_EPS, s), "norm1")
            self._linear(blk["qkv"], b["h"], b["qkv"], M, None, s)
            hip.check(lib.icamd_attention_fwd(b["qkv"].data_ptr(), b["ao"].data_ptr(), b["lse"].data_ptr(), B, T, self.heads, 64,
                                              scale, s), "attention")
            self._linear(blk["proj"], b["ao"], b["x1"], M, x, s)                  # x1 = x + proj(attn)
            hip.check(lib.icamd_layernorm_fwd(b["x1"].data_ptr(), self._pf(blk["n2w"]), self._pf(blk["n2b"]),
                                              b["h2"].data_ptr(), b["st2"].data_ptr(), b["st2"].data_ptr() + 4 * M, M, D, LN_EPS,
                                              s), "norm2")
            l1 = blk["fc1"]                                                       # z = fc1(h2), a = gelu(z): one kernel
            hip.check(lib.icamd_conv2d_fwd_gelu(ctypes.byref(l1.desc(M)), b["h2"].data_ptr(), self._w(l1), (None if logits_only else b["z"].data_ptr()),
                                                b["a"].data_ptr(), self._pf(l1.b), s), l1.name + " + gelu")
            self._linear(blk["fc2"], b["a"], b["x2"], M, b["x1"], s)              # x2 = x1 + mlp
            x = b["x2"]
        ws["x_last"] = x
        # final LayerNorm only on the class-token rows (the only rows the head reads)
        hip.check(lib.icamd_strided_rows_copy(x.data_ptr(), T * D, ws["cls_rows"].data_ptr(), D, B, D, s), "cls gather")
        hip.check(lib.icamd_layernorm_fwd(ws["cls_rows"].data_ptr(), self._pf(self.p_nw), self._pf(self.p_nb),
                                          ws["pooled"].data_ptr(), ws["stf"].data_ptr(), ws["stf"].data_ptr() + 4 * B, B, D,
                                          LN_EPS, s), "norm")
        self._linear(self.head, ws["pooled"], ws["logits"], B, None, s)
        return ws["logits"]

    def __call__(self, x_nchw):
        ws = self.pack(x_nchw.to(self.device, dtype=torch.float32).contiguous())
        return self.forward_packed(ws)[:, : self.num_classes]

    # ------------------------------------------------------------------ backward
    def backward_packed(self, ws, accumulate=False):
        lib, s = self.lib, hip.stream_ptr()
        B, M, D, T = ws["B"], ws["M"], self.dim, self.T
        acc = int(bool(accumulate))
        hook = self.grad_ready_hook
        wsp, wsb = ws["wg_ws"].data_ptr(), ws["wg_bytes"]
        lnp, lnb = ws["ln_ws"].data_ptr(), ws["ln_bytes"]
        csp, csb = ws["cs_ws"].data_ptr(), ws["cs_bytes"]
        scale = 64 ** -0.5

        lane = self._side_lane()
        lane.enabled = lane.side is not None and getattr(self, "wgrad_side_stream", True)
        lane.begin()

        def lin_bwd(l, x, dy, rows, dx, gelu_z=None):
            """weight, bias gradients (+ data gradient into dx when given) of y = x W^T + b.  The weight gradient goes to
            the side lane (streams.py); `dy` is protected from being overwritten until it has been read."""
            d = l.desc(rows)
            lane.launch(lambda st: hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(d), x.data_ptr(), dy.data_ptr(),
                                                                         self._gf(l.w), self._gf(l.b), acc, wsp, wsb, st),
                                             l.name + " wgrad+bias"), reads=(dy.data_ptr(),))
            if dx is not None:
                lane.before_write(dx.data_ptr())
                if gelu_z is None:
                    hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), self._wt(l), dx.data_ptr(), None, None, s),
                              l.name + " dgrad")
                else:   # dx = (dy W) * gelu'(z): the GELU backward rides in the data-gradient kernel's store pass
                    hip.check(lib.icamd_conv2d_dgrad_gelu(ctypes.byref(d), dy.data_ptr(), self._wt(l), gelu_z.data_ptr(),
                                                          dx.data_ptr(), s), l.name + " dgrad + gelu bwd")

        def ln_bwd(dy, x, st, wp, bp, addend, dx, rows):
            lane.before_write(dx.data_ptr())
            hip.check(lib.icamd_layernorm_bwd(dy.data_ptr(), x.data_ptr(), st.data_ptr(), st.data_ptr() + 4 * rows,
                                              self._pf(wp), None if addend is None else addend.data_ptr(), dx.data_ptr(),
                                              self._gf(wp), self._gf(bp), rows, D, acc, lnp, lnb, s), wp.name + " bwd")

        g0, g1, g2 = ws["g768"]
        lin_bwd(self.head, ws["pooled"], ws["dlogits"], B, ws["dpooled"])
        ln_bwd(ws["dpooled"], ws["cls_rows"], ws["stf"], self.p_nw, self.p_nb, None, ws["dcls"], B)
        if hook:
            hook(self.p_nw.offset, self.n_params, lane.events())
        dx = g0
        hip.check(lib.icamd_fill_zero(dx.data_ptr(), dx.numel() * 2, s), "zero")
        hip.check(lib.icamd_strided_rows_copy(ws["dcls"].data_ptr(), D, dx.data_ptr(), T * D, B, D, s), "cls scatter")
        spare = [g1, g2]
        for blk, b in zip(reversed(self.blocks), reversed(ws["blocks"])):
            # dx = grad wrt x2
            lin_bwd(blk["fc2"], b["a"], dx, M, ws["g3072b"], gelu_z=b["z"])       # d z = (dx W2) * gelu'(z)
            dh2 = spare[0]
            lin_bwd(blk["fc1"], b["h2"], ws["g3072b"], M, dh2)
            dx1 = spare[1]
            ln_bwd(dh2, b["x1"], b["st2"], blk["n2w"], blk["n2b"], dx, dx1, M)    # dx1 = LN2'(dh2) + dx
            dao = dh2
            lin_bwd(blk["proj"], b["ao"], dx1, M, dao)
            lane.before_write(ws["g2304"].data_ptr())
            hip.check(lib.icamd_attention_bwd(b["qkv"].data_ptr(), b["ao"].data_ptr(), dao.data_ptr(), b["lse"].data_ptr(),
                                              ws["delta"].data_ptr(), ws["g2304"].data_ptr(), B, T, self.heads, 64, scale, s),
                      "attention bwd")
            dh = dao
            lin_bwd(blk["qkv"], b["h"], ws["g2304"], M, dh)
            dxin = dx                                                             # reuse: dx is dead after ln2 bwd
            ln_bwd(dh, b["x"], b["st1"], blk["n1w"], blk["n1b"], dx1, dxin, M)    # dx_in = LN1'(dh) + dx1
            dx = dxin
            if hook:
                hook(blk["n1w"].offset, None, lane.events())
        # tokens -> cls_token, pos_embed, patches
        hip.check(lib.icamd_batch_sum(dx.data_ptr(), T * D, B, T * D, self._gf(self.p_pos), acc, s), "pos_embed grad")
        hip.check(lib.icamd_batch_sum(dx.data_ptr(), T * D, B, D, self._gf(self.p_cls), acc, s), "cls_token grad")
        dpatch = ws["patches"]   # forward value no longer needed
        hip.check(lib.icamd_strided_rows_copy(dx.data_ptr() + 2 * D, T * D, dpatch.data_ptr(), (T - 1) * D, B, (T - 1) * D, s),
                  "patch grads")
        dpe = self._pe_desc(B)
        lane.launch(lambda st: hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(dpe), ws["x8"].data_ptr(), dpatch.data_ptr(),
                                                                     self._gf(self.pe_w), self._gf(self.pe_b), acc, wsp, wsb,
                                                                     st), "patch_embed wgrad+bias"))
        lane.join()
        if hook:
            hook(0, None)

    def _side_lane(self):
        if getattr(self, "_lane", None) is None:
            from .streams import SideLane
            # opt-in for ViT: its main-stream chain is itself MFMA-bound GEMMs + attention, so concurrent weight-gradient
            # GEMMs only compete with it (measured 54.1 -> 54.7 ms/step at batch 256); the CNNs default to on
            self._lane = SideLane(self.device, os.environ.get("ICAMD_WGRAD_STREAM_VIT", "0") == "1")
        return self._lane
