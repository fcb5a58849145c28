"""ConvNeXt (timm `convnext_tiny` family) on the gfx950 kernels: hand-written forward and backward.

BASELINE.json configs[4] is ConvNeXt-T + mixup/cutmix + EMA.  The block is specified in the reference tree
(/root/reference/semantic_segmentation/backbone/convnext.py:21-56: dwconv 7x7 -> LayerNorm -> Linear 4x -> GELU ->
Linear -> gamma -> drop_path -> residual; stem 4x4/4 conv + channels-first LN :79-82; downsample LN + 2x2/2 conv
:85-88; layer-scale init 1e-6 :32,39); the classification model itself comes from timm (train.py:194), whose parameter
names are used here: `stem.{0,1}`, `stages.S.downsample.{0,1}`, `stages.S.blocks.B.{conv_dw,norm,mlp.fc1,mlp.fc2,gamma}`,
`head.{norm,fc}`.  Stochastic depth follows timm: linearly increasing rates up to `drop_path_rate`
(train.py:189-192 passes --drop_path, default 0.05), per-sample masks drawn on the host from torch's RNG.

NHWC activations make every "channels-first LayerNorm" an ordinary row LayerNorm over pixels, and every Linear a 1x1
convolution on the same tensor (bias fused); the depthwise stencil and the layer-scale tail are dedicated kernels.
"""
import ctypes
import os
from collections import OrderedDict

import torch

from . import hip
from .checkpoint import PicklableModel
from .vit import _P, _align

LN_EPS = 1e-6
# Round 5: the layer scale (and timm drop_path's 1 / keep_prob) folded into fc2's filter and bias, the residual added in fc2's store
# pass (icamd_layerscale_fold; DESIGN.md section 5, round-5 finding 10).  0: fc2 -> icamd_layerscale_fwd / _bwd as in rounds 2-4.
_FUSED_LS = os.environ.get("ICAMD_FUSED_LAYERSCALE", "1") != "0"

CONFIGS = {
    "convnext_tiny": ((3, 3, 9, 3), (96, 192, 384, 768)),
    "convnext_small": ((3, 3, 27, 3), (96, 192, 384, 768)),
    "convnext_test": ((1, 1, 2, 1), (32, 64, 128, 192)),   # small configuration for parity tests
    # the backbone of tests/golden/convnext_ref_vectors.npz (vectors from the reference's own ConvNeXt class,
    # /root/reference/semantic_segmentation/backbone/convnext.py:58-150)
    "convnext_pin": ((1, 1, 1, 1), (32, 64, 96, 192)),
}


class _Conv:
    def __init__(self, name, cin, cout, k, stride, cin_p=None, cout_p=None):
        self.name, self.cin, self.cout, self.k, self.stride = name, cin, cout, k, stride
        self.cin_p, self.cout_p = cin_p or cin, cout_p or cout
        self.w = self.b = None
        self.wt_offset = None
        self.descs = {}

    def desc(self, N, H, W):
        key = (N, H, W)
        d = self.descs.get(key)
        if d is None:
            d = hip.conv_desc(N, H, W, self.cin_p, self.cout_p, self.k, self.k, self.stride, 0)
            self.descs[key] = d
        return d


class ConvNeXt(PicklableModel):
    def __init__(self, arch="convnext_tiny", num_classes=1000, device="cuda", drop_path_rate=0.0, seed=None):
        hip.require_gpu()
        self.lib = hip.load()
        self.arch, self.num_classes = arch, num_classes
        self.device = torch.device(device)
        self.training = True
        self.depths, self.dims = CONFIGS[arch]
        self.drop_path_rate = drop_path_rate
        self.ncls_p = _align(num_classes, 64)
        self.num_batches_tracked = 0
        self.grad_ready_hook = None
        self.injected_keep = None     # tests: list of per-block keep tensors (float [B]) to use instead of drawing
        self._ws = {}
        self._build()
        self.init_weights(seed)

    # ------------------------------------------------------------------ structure / arenas
    def _build(self):
        dev = self.device
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

        self.gemms = []   # layers that need a transposed shadow (data gradient)

        def conv(name, cin, cout, k, stride, cin_p=None, cout_p=None, needs_dgrad=True, lin=False):
            c = _Conv(name, cin, cout, k, stride, cin_p, cout_p)
            if lin:
                c.w = add(name + ".weight", (cout, cin), "lin", (c.cout_p, cin))
            else:
                c.w = add(name + ".weight", (cout, cin, k, k), "conv", (c.cout_p, k, k, c.cin_p))
            c.b = add(name + ".bias", (cout,), "vec", (c.cout_p,))
            if needs_dgrad:
                self.gemms.append(c)
            return c

        d0 = self.dims[0]
        self.stem = conv("stem.0", 3, d0, 4, 4, cin_p=8, needs_dgrad=False)
        self.stem_nw = add("stem.1.weight", (d0,), "vec", (d0,))
        self.stem_nb = add("stem.1.bias", (d0,), "vec", (d0,))
        self.stages = []
        rates = torch.linspace(0, self.drop_path_rate, sum(self.depths)).tolist()
        bi = 0
        for si, (depth, dim) in enumerate(zip(self.depths, self.dims)):
            st = {"dim": dim, "blocks": []}
            if si > 0:
                prev = self.dims[si - 1]
                st["ds_nw"] = add(f"stages.{si}.downsample.0.weight", (prev,), "vec", (prev,))
                st["ds_nb"] = add(f"stages.{si}.downsample.0.bias", (prev,), "vec", (prev,))
                st["ds"] = conv(f"stages.{si}.downsample.1", prev, dim, 2, 2)
            for j in range(depth):
                n = f"stages.{si}.blocks.{j}"
                blk = {"name": n, "rate": rates[bi]}
                blk["dw_w"] = add(f"{n}.conv_dw.weight", (dim, 1, 7, 7), "dw", (7, 7, dim))
                blk["dw_b"] = add(f"{n}.conv_dw.bias", (dim,), "vec", (dim,))
                blk["nw"] = add(f"{n}.norm.weight", (dim,), "vec", (dim,))
                blk["nb"] = add(f"{n}.norm.bias", (dim,), "vec", (dim,))
                blk["fc1"] = conv(f"{n}.mlp.fc1", dim, 4 * dim, 1, 1, lin=True)
                blk["fc2"] = conv(f"{n}.mlp.fc2", 4 * dim, dim, 1, 1, lin=True)
                blk["gamma"] = add(f"{n}.gamma", (dim,), "vec", (dim,))
                st["blocks"].append(blk)
                bi += 1
            self.stages.append(st)
        dl = self.dims[-1]
        self.head_nw = add("head.norm.weight", (dl,), "vec", (dl,))
        self.head_nb = add("head.norm.bias", (dl,), "vec", (dl,))
        self.head = conv("head.fc", dl, self.num_classes, 1, 1, cout_p=self.ncls_p, lin=True)
        self.n_params = off
        self.param_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.buffer_arena = torch.zeros(64, dtype=torch.float32, device=dev)
        toff, descs, tjobs, jobs = 0, [], [], []
        for c in self.gemms:
            c.wt_offset = toff
            T = c.k * c.k
            descs.append([c.w.offset, toff, c.cout_p, T, c.cin_p, 0, 0, 0])
            if c.cout_p % 64 == 0 and c.cin_p % 64 == 0:
                tjobs += [[len(descs) - 1, t, a, b] for t in range(T) for a in range(0, c.cout_p, 64)
                          for b in range(0, c.cin_p, 64)]
            else:
                jobs += [[len(descs) - 1, s] for s in range(0, c.w.numel, 4096)]
            toff = _align(toff + c.w.numel, 128)
        self.shadow_t = torch.zeros(toff, dtype=torch.bfloat16, device=dev)
        self._tr_descs = torch.tensor(descs, dtype=torch.int64, device=dev)
        self._tr_tjobs = torch.tensor(tjobs if tjobs else [[0, 0, 0, 0]], dtype=torch.int32, device=dev)
        self._tr_ntjobs = len(tjobs)
        self._tr_jobs = torch.tensor(jobs if jobs else [[0, 0]], dtype=torch.int32, device=dev)
        self._tr_njobs = len(jobs)
        # folded layer scale: one folded fc2 bias per block; the folded filter takes fc2's place in the bf16 shadow (and, through
        # refresh_transposed, in the transposed shadow), so _w(fc2) / _wt(fc2) are the folded operands
        self.fused_ls = _FUSED_LS
        fb = 0
        for st in self.stages:
            for blk in st["blocks"]:
                blk["fb_off"] = fb
                fb = _align(fb + blk["fc2"].cout_p, 64)
        self.fold_bias = torch.zeros(max(fb, 64), dtype=torch.float32, device=dev)
        self._ls_cbs = None          # the per-block constants the shadow is currently folded with
        self._ls_jobs = None

    def _ctor_kwargs(self):
        return {"arch": self.arch, "num_classes": self.num_classes, "drop_path_rate": self.drop_path_rate}

    def init_weights(self, seed=None):
        """timm ConvNeXt init: trunc_normal(std .02) conv / linear weights, zero biases, LayerNorm 1 / 0, gamma 1e-6."""
        g = torch.Generator()
        g.manual_seed(seed if seed is not None else torch.initial_seed() % (2 ** 63))
        sd = OrderedDict()
        for name, p in self.params.items():
            if p.kind in ("conv", "lin", "dw"):
                sd[name] = torch.nn.init.trunc_normal_(torch.empty(p.torch_shape), std=0.02, generator=g)
            elif name.endswith(".gamma"):
                sd[name] = torch.full(p.torch_shape, 1e-6)
            elif name.endswith("weight"):
                sd[name] = torch.ones(p.torch_shape)
            else:
                sd[name] = torch.zeros(p.torch_shape)
        self.load_state_dict(sd)

    def _to_arena(self, p, t):
        t = t.detach().to(torch.float32).cpu()
        full = torch.zeros(p.padded_shape)
        if p.kind == "conv":
            full[: t.shape[0], :, :, : t.shape[1]] = t.permute(0, 2, 3, 1)
        elif p.kind == "lin":
            full[: t.shape[0], :] = t.reshape(t.shape[0], -1)
        elif p.kind == "dw":
            full[:] = t.reshape(t.shape[0], 7, 7).permute(1, 2, 0)
        else:
            full.view(-1)[: t.numel()] = t.flatten()
        return full.flatten()

    def _from_arena(self, p, flat):
        t = flat.reshape(p.padded_shape)
        if p.kind == "conv":
            return t[: p.torch_shape[0], :, :, : p.torch_shape[1]].permute(0, 3, 1, 2).contiguous()
        if p.kind == "lin":
            return t[: p.torch_shape[0], :].clone()
        if p.kind == "dw":
            return t.permute(2, 0, 1).reshape(p.torch_shape).contiguous()
        return t.flatten()[: p.torch_shape[0]].clone()

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

    def _ls_mode_cbs(self):
        """timm drop_path scales the kept samples by 1 / keep_prob in training; identity in eval."""
        blocks = [blk for st in self.stages for blk in st["blocks"]]
        if self.training and self.injected_keep is not None:      # tests: the scale is whatever the injected masks carry
            cbs = []
            for blk, k in zip(blocks, self.injected_keep):
                k = k.float().cpu()
                cb = float(k.max()) if float(k.max()) > 0.0 else 1.0
                assert bool(((k == 0) | (k == cb)).all()), "folded layer scale: a keep mask is {0, c} per block"
                cbs.append(cb if blk["rate"] > 0.0 else 1.0)
            return cbs
        return [1.0 / (1.0 - blk["rate"]) if (self.training and blk["rate"] > 0.0) else 1.0 for blk in blocks]

    def _fold_layerscale(self):
        import struct
        cbs = self._ls_mode_cbs()
        if self._ls_jobs is None or cbs != self._ls_cbs:
            rows, row0, elems = [], 0, 0
            blocks = [blk for st in self.stages for blk in st["blocks"]]
            for blk, cb in zip(blocks, cbs):
                c = blk["fc2"]
                bits = struct.unpack("<i", struct.pack("<f", cb))[0]
                rows.append([c.w.offset, blk["gamma"].offset, c.b.offset, blk["fb_off"], c.cout, c.cin, row0, bits])
                row0 += c.cout
                elems += c.cout * c.cin
            self._ls_jobs = torch.tensor(rows, dtype=torch.int64, device=self.device)
            self._ls_rows, self._ls_elems, self._ls_cbs = row0, elems, cbs
        hip.check(self.lib.icamd_layerscale_fold(self.param_arena.data_ptr(), self.shadow.data_ptr(), self.fold_bias.data_ptr(),
                                                 self._ls_jobs.data_ptr(), self._ls_jobs.shape[0], self._ls_rows, self._ls_elems,
                                                 hip.stream_ptr()), "layer scale fold")

    def refresh_transposed(self):
        s = hip.stream_ptr()
        if self.fused_ls:
            self._fold_layerscale()      # before the transposes: fc2's slot of the shadow becomes the folded filter
        if self._tr_ntjobs:
            hip.check(self.lib.icamd_filter_transpose_tiled(self.shadow.data_ptr(), self.shadow_t.data_ptr(),
                                                            self._tr_descs.data_ptr(), self._tr_tjobs.data_ptr(),
                                                            self._tr_ntjobs, s), "filter_transpose_tiled")
        if self._tr_njobs:
            hip.check(self.lib.icamd_filter_transpose(self.shadow.data_ptr(), self.shadow_t.data_ptr(),
                                                      self._tr_descs.data_ptr(), self._tr_jobs.data_ptr(), self._tr_njobs, s),
                      "filter_transpose")

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    # ------------------------------------------------------------------ workspace
    def _workspace(self, N, H, W):
        key = (N, H, W)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, lib = self.device, self.lib

        def act(*shape):
            return torch.empty(*shape, dtype=torch.bfloat16, device=dev)

        def f32(n):
            return torch.empty(n, dtype=torch.float32, device=dev)

        ws = {"N": N, "H": H, "W": W}
        ws["x8"] = act(N, H, W, 8)
        h, w = H // 4, W // 4
        d0 = self.dims[0]
        ws["s"] = act(N, h, w, d0)
        ws["x0"] = act(N, h, w, d0)
        ws["st_stem"] = f32(2 * N * h * w)
        max_act, max_rows = N * h * w * 4 * d0, N * h * w
        wg = lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(self.stem.desc(N, H, W)))
        dwg = 0
        stages = []
        for si, st in enumerate(self.stages):
            dim = st["dim"]
            sw = {"blocks": []}
            if si > 0:
                sw["ln"] = act(N, h, w, self.dims[si - 1])
                sw["st"] = f32(2 * N * h * w)
                wg = max(wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(st["ds"].desc(N, h, w))))
                h, w = h // 2, w // 2
                sw["x"] = act(N, h, w, dim)
            sw["hw"] = (h, w)
            rows = N * h * w
            for blk in st["blocks"]:
                sw["blocks"].append({"d": act(N, h, w, dim), "h": act(N, h, w, dim), "z1": act(N, h, w, 4 * dim),
                                     "a": act(N, h, w, 4 * dim), "z2": None if self.fused_ls else act(N, h, w, dim), "out": act(N, h, w, dim),
                                     "st": f32(2 * rows), "keep": None})
                wg = max(wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(blk["fc1"].desc(N, h, w))),
                         lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(blk["fc2"].desc(N, h, w))))
                dwg = max(dwg, lib.icamd_dwconv7_wgrad_workspace_bytes(N, h, w, dim))
                max_act = max(max_act, rows * 4 * dim)
            stages.append(sw)
        ws["stages"] = stages
        ws["final_hw"] = (h, w)
        dl = self.dims[-1]
        ws["pool"] = act(N, dl)
        ws["pn"] = act(N, dl)
        ws["st_head"] = f32(2 * N)
        ws["logits"] = torch.zeros(N, self.ncls_p, dtype=torch.bfloat16, device=dev)
        ws["dlogits"] = torch.zeros(N, self.ncls_p, dtype=torch.bfloat16, device=dev)
        ws["loss_rows"] = f32(N)
        ws["pred"] = torch.empty(N, dtype=torch.int32, device=dev)
        wg = max(wg, lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(self.head.desc(N, 1, 1))))
        ws["wg_ws"] = torch.empty(wg, dtype=torch.uint8, device=dev)
        ws["wg_bytes"] = wg
        ws["dwg_ws"] = torch.empty(max(dwg, 256), dtype=torch.uint8, device=dev)
        ws["dwg_bytes"] = dwg
        maxc = 4 * max(self.dims)
        ws["ln_bytes"] = max(lib.icamd_layernorm_bwd_workspace_bytes(max_rows, c) for c in self.dims)
        ws["ln_ws"] = torch.zeros(ws["ln_bytes"], dtype=torch.uint8, device=dev)
        ws["cs_bytes"] = max(lib.icamd_colsum_rows_workspace_bytes(max_rows, min(maxc, 4096)),
                             lib.icamd_layerscale_bwd_workspace_bytes(max_rows, max(self.dims)))
        ws["cs_ws"] = torch.zeros(ws["cs_bytes"], dtype=torch.uint8, device=dev)
        ws["max_act"] = max_act
        if self.fused_ls:     # side-lane scratch of the folded layer's parameter gradients
            dmax = max(self.dims)
            ws["ls_G"] = f32(4 * dmax * dmax)
            ws["ls_S"] = f32(dmax)
            ws["ls_drop"] = f32(N * dmax)
        self._ws[key] = ws
        return ws

    def _scratch(self, ws):
        if "g" not in ws:
            ws["g"] = [torch.empty(ws["max_act"], dtype=torch.bfloat16, device=self.device) for _ in range(5)]
        return ws["g"]

    # ------------------------------------------------------------------ helpers
    def _pf(self, p):
        return self.param_arena.data_ptr() + 4 * p.offset

    def _gf(self, p):
        return self.grad_arena.data_ptr() + 4 * p.offset

    def _w(self, c):
        return self.shadow.data_ptr() + 2 * c.w.offset

    def _wt(self, c):
        return self.shadow_t.data_ptr() + 2 * c.wt_offset

    def pack(self, x_nchw, mix=None):
        N, C, H, W = x_nchw.shape
        ws = self._workspace(N, H, W)
        mode, lam, box = (0, 1.0, (0, 0, 0, 0)) if mix is None else mix
        hip.check(self.lib.icamd_pack_input(x_nchw.data_ptr(), ws["x8"].data_ptr(), N, C, H, W, mode, float(lam), int(box[0]),
                                            int(box[1]), int(box[2]), int(box[3]), hip.stream_ptr()), "pack")
        return ws

    def _conv(self, c, x_ptr, y, N, H, W, s):
        hip.check(self.lib.icamd_conv2d_fwd(ctypes.byref(c.desc(N, H, W)), x_ptr, self._w(c), y.data_ptr(), self._pf(c.b), None,
                                            None, s), c.name)

    def _ln(self, x, wp, bp, y, st, rows, C, s):
        hip.check(self.lib.icamd_layernorm_fwd(x.data_ptr(), self._pf(wp), self._pf(bp), y.data_ptr(), st.data_ptr(),
                                               st.data_ptr() + 4 * rows, rows, C, LN_EPS, s), wp.name)

    # ------------------------------------------------------------------ forward
    def forward_packed(self, ws, logits_only=False):
        """logits_only: a forward whose activations no backward will read (the reference's second, accuracy-only forward under
        mixup): tensors kept only for the backward pass (the pre-GELU Mlp activations) are not written."""
        lib, s = self.lib, hip.stream_ptr()
        N, H, W = ws["N"], ws["H"], ws["W"]
        self._conv(self.stem, ws["x8"].data_ptr(), ws["s"], N, H, W, s)
        h, w = H // 4, W // 4
        self._ln(ws["s"], self.stem_nw, self.stem_nb, ws["x0"], ws["st_stem"], N * h * w, self.dims[0], s)
        x = ws["x0"]
        bi = 0
        # stochastic depth (timm drop_path: per sample, scaled by 1/keep_prob): every block's mask for this step is drawn in
        # ONE host call and uploaded ONCE from pinned memory without blocking, so the host keeps running ahead of the device
        drop_rows = None
        if self.training and self.injected_keep is None:
            rates = [blk["rate"] for st in self.stages for blk in st["blocks"]]
            if any(r > 0.0 for r in rates):
                kp = 1.0 - torch.tensor(rates, dtype=torch.float32).view(-1, 1)
                # a ring of 4 pinned staging rows: the upload of step i is enqueued behind step i's stem kernels, so waiting for it
                # before the NEXT draw (one buffer, rounds 2-3) tied the host to within one step of the device (12 ms of the host's
                # step spent in Event.synchronize, round-4 profile); with four rows the wait is for the upload issued 4 steps ago
                ring = ws.get("keep_host")
                if ring is None or ring.shape[1:] != (len(rates), N):
                    ring = ws["keep_host"] = torch.empty(4, len(rates), N, dtype=torch.float32).pin_memory()
                    ws["keep_dev"] = torch.empty(len(rates), N, dtype=torch.float32, device=self.device)
                    ws["keep_copied"] = [None] * 4
                    ws["keep_slot"] = 0
                slot = ws["keep_slot"]
                ws["keep_slot"] = (slot + 1) % 4
                host = ring[slot]
                if ws["keep_copied"][slot] is not None:
                    ws["keep_copied"][slot].synchronize()   # the upload that last used this row has left it
                torch.div((torch.rand(len(rates), N) < kp).float(), kp, out=host)
                ws["keep_dev"].copy_(host, non_blocking=True)
                ws["keep_copied"][slot] = torch.cuda.Event()
                ws["keep_copied"][slot].record()
                drop_rows = ws["keep_dev"]
        if self.fused_ls and self._ls_cbs != self._ls_mode_cbs():
            self.refresh_transposed()     # train() <-> eval(), or a test's injected masks: fold with this mode's constants
        for si, (st, sw) in enumerate(zip(self.stages, ws["stages"])):
            dim = st["dim"]
            if si > 0:
                sw["in"] = x
                self._ln(x, st["ds_nw"], st["ds_nb"], sw["ln"], sw["st"], N * h * w, self.dims[si - 1], s)
                self._conv(st["ds"], sw["ln"].data_ptr(), sw["x"], N, h, w, s)
                h, w = h // 2, w // 2
                x = sw["x"]
            rows = N * h * w
            for blk, b in zip(st["blocks"], sw["blocks"]):
                b["in"] = x
                hip.check(lib.icamd_dwconv7_fwd(x.data_ptr(), self.shadow.data_ptr() + 2 * blk["dw_w"].offset,
                                                self._pf(blk["dw_b"]), b["d"].data_ptr(), N, h, w, dim, s), blk["name"] + " dw")
                self._ln(b["d"], blk["nw"], blk["nb"], b["h"], b["st"], rows, dim, s)
                c1 = blk["fc1"]                                                   # z1 = pwconv1(h), a = gelu(z1): one kernel
                hip.check(lib.icamd_conv2d_fwd_gelu(ctypes.byref(c1.desc(N, h, w)), b["h"].data_ptr(), self._w(c1),
                                                    (None if logits_only else b["z1"].data_ptr()), b["a"].data_ptr(), self._pf(c1.b), s), c1.name + " + gelu")
                keep = None
                if self.training and blk["rate"] > 0.0:
                    if self.injected_keep is not None:
                        keep = self.injected_keep[bi].to(self.device, dtype=torch.float32)
                    else:
                        keep = drop_rows[bi]
                b["keep"] = keep
                if self.fused_ls:
                    # out = x + a W2'^T + b2' in fc2's store pass; the dropped samples: out = x, and their rows of `a` are cleared
                    # so that they vanish from fc2's weight gradient (which then runs on the unmasked output gradient)
                    c2 = blk["fc2"]
                    hip.check(lib.icamd_conv2d_fwd(ctypes.byref(c2.desc(N, h, w)), b["a"].data_ptr(), self._w(c2),
                                                   b["out"].data_ptr(), self.fold_bias.data_ptr() + 4 * blk["fb_off"],
                                                   x.data_ptr(), None, s), c2.name + " + layer scale + residual")
                    if keep is not None:
                        hip.check(lib.icamd_rows_fix(keep.data_ptr(), N, b["out"].data_ptr(), x.data_ptr(), h * w * dim * 2,
                                                     None if logits_only else b["a"].data_ptr(), h * w * dim * 8, s), "drop path")
                else:
                    self._conv(blk["fc2"], b["a"].data_ptr(), b["z2"], N, h, w, s)
                    hip.check(lib.icamd_layerscale_fwd(b["z2"].data_ptr(), x.data_ptr(), self._pf(blk["gamma"]),
                                                       None if keep is None else keep.data_ptr(), b["out"].data_ptr(), rows, dim,
                                                       h * w, s), "layer scale")
                x = b["out"]
                bi += 1
        dl = self.dims[-1]
        hip.check(lib.icamd_avgpool_fwd(x.data_ptr(), ws["pool"].data_ptr(), N, h * w, dl, s), "avgpool")
        self._ln(ws["pool"], self.head_nw, self.head_nb, ws["pn"], ws["st_head"], N, dl, s)
        self._conv(self.head, ws["pn"].data_ptr(), ws["logits"], N, 1, 1, s)
        return ws["logits"]

    def __call__(self, x_nchw):
        ws = self.pack(x_nchw.to(self.device, dtype=torch.float32).contiguous())
        return self.forward_packed(ws)[:, : self.num_classes]

    # ------------------------------------------------------------------ backward
    def backward_packed(self, ws, accumulate=False, dfeat=None):
        """Backward from ws['dlogits'].  `dfeat` (bf16 NHWC tensor shaped like the last stage's output): start from that
        gradient instead and skip the classification head -- the entry the reference-vector parity test uses, since the
        reference tree's ConvNeXt is the headless backbone."""
        lib, s = self.lib, hip.stream_ptr()
        N = ws["N"]
        acc = int(bool(accumulate))
        hook = self.grad_ready_hook
        wsp, wsb = ws["wg_ws"].data_ptr(), ws["wg_bytes"]
        dwp, dwb = ws["dwg_ws"].data_ptr(), ws["dwg_bytes"]
        lnp, lnb = ws["ln_ws"].data_ptr(), ws["ln_bytes"]
        csp, csb = ws["cs_ws"].data_ptr(), ws["cs_bytes"]
        G = [g.data_ptr() for g in self._scratch(ws)]

        lane = self._side_lane()
        lane.enabled = lane.side is not None and getattr(self, "wgrad_side_stream", True)
        lane.begin()

        def W(ptr):
            """`ptr` is about to be overwritten on the main stream: wait for side-lane launches still reading it."""
            lane.before_write(ptr)
            return ptr

        def gemm_bwd(c, x_ptr, dy_ptr, n, h, w, dx_ptr, gelu_z=None):
            d = c.desc(n, h, w)
            lane.launch(lambda st_: hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(d), x_ptr, dy_ptr, self._gf(c.w),
                                                                          self._gf(c.b), acc, wsp, wsb, st_),
                                              c.name + " wgrad+bias"), reads=(dy_ptr,))
            if dx_ptr is not None and gelu_z is None:
                hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy_ptr, self._wt(c), W(dx_ptr), None, None, s), c.name + " dgrad")
            elif dx_ptr is not None:   # dx = (dy W) * gelu'(z) in the data-gradient kernel's store pass
                hip.check(lib.icamd_conv2d_dgrad_gelu(ctypes.byref(d), dy_ptr, self._wt(c), gelu_z, W(dx_ptr), s),
                          c.name + " dgrad + gelu bwd")

        def ln_bwd(dy_ptr, x, st, wp, bp, dx_ptr, rows, C):
            hip.check(lib.icamd_layernorm_bwd(dy_ptr, x.data_ptr(), st.data_ptr(), st.data_ptr() + 4 * rows, self._pf(wp), None,
                                              W(dx_ptr), self._gf(wp), self._gf(bp), rows, C, acc, lnp, lnb, s), wp.name + " bwd")

        dl = self.dims[-1]
        h, w = ws["final_hw"]
        dout = G[0]
        if dfeat is None:
            gemm_bwd(self.head, ws["pn"].data_ptr(), ws["dlogits"].data_ptr(), N, 1, 1, G[1])
            ln_bwd(G[1], ws["pool"], ws["st_head"], self.head_nw, self.head_nb, G[2], N, dl)
            hip.check(lib.icamd_avgpool_bwd(G[2], W(dout), N, h * w, dl, s), "avgpool bwd")
        else:
            assert dfeat.dtype == torch.bfloat16 and dfeat.numel() == N * h * w * dl
            self._scratch(ws)[0][: dfeat.numel()].copy_(dfeat.reshape(-1))
        if hook:
            hook(self.head_nw.offset, self.n_params, lane.events())
        other = G[3]
        nblocks, bj = sum(self.depths), 0     # bj: blocks done, counted from the last one
        for si in range(len(self.stages) - 1, -1, -1):
            st, sw = self.stages[si], ws["stages"][si]
            dim = st["dim"]
            rows = N * h * w
            for blk, b in zip(reversed(st["blocks"]), reversed(sw["blocks"])):
                keep = None if b["keep"] is None else b["keep"].data_ptr()
                if self.fused_ls:
                    # fc2 with the layer scale folded in: its data gradient reads the block's output gradient itself (the folded,
                    # transposed filter carries cb * gamma); the side lane forms G = dout^T a and the column sums of dout over the
                    # kept samples, and from them the gradients of W2, b2 and gamma (icamd_layerscale_param_grads)
                    c2, gam = blk["fc2"], blk["gamma"]
                    d2 = c2.desc(N, h, w)
                    cb = self._ls_cbs[nblocks - 1 - bj]
                    lsG, lsS, lsD = ws["ls_G"].data_ptr(), ws["ls_S"].data_ptr(), ws["ls_drop"].data_ptr()

                    def side(st_, d2=d2, c2=c2, gam=gam, cb=cb, keep=keep, a_ptr=b["a"].data_ptr(), dout=dout, hw=h * w, dim=dim):
                        if keep is not None:
                            hip.check(lib.icamd_dropped_colsum(dout, keep, N, hw, dim, lsD, st_), "dropped column sums")
                        hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(d2), a_ptr, dout, lsG, lsS, 0, wsp, wsb, st_),
                                  c2.name + " wgrad+bias (folded)")
                        hip.check(lib.icamd_layerscale_param_grads(lsG, self._pf(c2.w), self._pf(c2.b), self._pf(gam), lsS,
                                                                   None if keep is None else lsD, N, cb, c2.cout, c2.cin,
                                                                   self._gf(c2.w), self._gf(c2.b), self._gf(gam), acc, st_),
                                  "layer scale parameter gradients")
                    lane.launch(side, reads=(dout,))
                    hip.check(lib.icamd_conv2d_dgrad_gelu(ctypes.byref(d2), dout, self._wt(c2), b["z1"].data_ptr(), W(G[4]), s),
                              c2.name + " dgrad + gelu bwd (folded)")
                    if keep is not None:
                        hip.check(lib.icamd_rows_fix(keep, N, G[4], None, h * w * dim * 8, None, 0, s), "drop path bwd")
                    bj += 1
                else:
                    hip.check(lib.icamd_layerscale_bwd(dout, b["z2"].data_ptr(), self._pf(blk["gamma"]), keep, W(G[1]),
                                                       self._gf(blk["gamma"]), rows, dim, h * w, acc, csp, csb, s), "layer scale bwd")
                    gemm_bwd(blk["fc2"], b["a"].data_ptr(), G[1], N, h, w, G[4], gelu_z=b["z1"].data_ptr())   # G4 = d z1
                gemm_bwd(blk["fc1"], b["h"].data_ptr(), G[4], N, h, w, G[1])                 # G1 = d h
                ln_bwd(G[1], b["d"], b["st"], blk["nw"], blk["nb"], G[2], rows, dim)         # G2 = d (dwconv out)
                bin_ptr, g2, hh, ww_, dd = b["in"].data_ptr(), G[2], h, w, dim
                dwg = self._gf(blk["dw_w"])
                if lib.icamd_dwconv7_wgrad_bias_supported(N, hh, ww_, dd):   # filter and bias gradient out of one pass over dy
                    dbg = self._gf(blk["dw_b"])
                    lane.launch(lambda st_, bin_ptr=bin_ptr, g2=g2, hh=hh, ww_=ww_, dd=dd, dwg=dwg, dbg=dbg, nm=blk["name"]: hip.check(
                        lib.icamd_dwconv7_wgrad_bias(bin_ptr, g2, dwg, dbg, acc, dwp, dwb, N, hh, ww_, dd, st_), nm + " dw wgrad+bias"),
                        reads=(G[2],))
                else:
                    lane.launch(lambda st_, bin_ptr=bin_ptr, g2=g2, hh=hh, ww_=ww_, dd=dd, dwg=dwg, nm=blk["name"]: hip.check(
                        lib.icamd_dwconv7_wgrad(bin_ptr, g2, dwg, acc, dwp, dwb, N, hh, ww_, dd, st_), nm + " dw wgrad"),
                        reads=(G[2],))
                    hip.check(lib.icamd_colsum_rows(G[2], rows, dim, dim, self._gf(blk["dw_b"]), acc, csp, csb, s), "dw bias grad")
                hip.check(lib.icamd_dwconv7_dgrad(G[2], self.shadow.data_ptr() + 2 * blk["dw_w"].offset, dout, W(other), N, h, w,
                                                  dim, s), blk["name"] + " dw dgrad")        # + residual gradient
                dout, other = other, dout
                if hook:
                    hook(blk["dw_w"].offset, None, lane.events())
            if si > 0:
                prev = self.dims[si - 1]
                gemm_bwd(st["ds"], sw["ln"].data_ptr(), dout, N, 2 * h, 2 * w, G[1])
                h, w = 2 * h, 2 * w
                ln_bwd(G[1], sw["in"], sw["st"], st["ds_nw"], st["ds_nb"], other, N * h * w, prev)
                dout, other = other, dout
                if hook:
                    hook(st["ds_nw"].offset, None, lane.events())
        ln_bwd(dout, ws["s"], ws["st_stem"], self.stem_nw, self.stem_nb, G[1], N * h * w, self.dims[0])
        gemm_bwd(self.stem, ws["x8"].data_ptr(), G[1], N, ws["H"], ws["W"], None)
        lane.join()
        if hook:
            hook(0, None)

    def _side_lane(self):
        if getattr(self, "_lane", None) is None:
            from .streams import SideLane
            self._lane = SideLane(self.device, os.environ.get("ICAMD_WGRAD_STREAM", "1") != "0")
        return self._lane
