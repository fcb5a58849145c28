"""Generates tests/golden/convnext_ref_vectors.npz by executing the REFERENCE's own ConvNeXt definition on CPU.

Run in the build container only (needs /root/reference; only the .npz travels):
    python tests/golden/make_convnext_fixture.py

What runs: /root/reference/semantic_segmentation/backbone/convnext.py, imported unmodified -- `Block` (:21-56),
`LayerNorm` (:158-182, both data formats) and the `ConvNeXt` backbone (:58-150: stem :79-82, downsample layers :85-88,
stages :91-99, forward_features :138-150).  Its import-time dependencies that are not installed (timm.models.layers,
mmcv_custom, mmseg) are satisfied by in-memory stand-ins that carry NO arithmetic on the path exercised here:
`trunc_normal_` = torch.nn.init.trunc_normal_ (initialisation only; every parameter is overwritten below with recorded
values), `DropPath` is never constructed (drop_path = 0 -> nn.Identity, convnext.py:40), `load_checkpoint` /
`get_root_logger` are never called, and `BACKBONES.register_module()` returns the class unchanged.

Vectors (all fp32 CPU arithmetic of the reference code; inputs and parameters are bf16-representable so that the HIP path
can consume them without a rounding of its own):
  net/*    the backbone with depths (1,1,1,1), dims (32,64,96,192), layer scale ~1: input [2,3,64,64], the raw output of
           every stage, and d<last stage output, r>/d(every parameter, input)
  blk96/*, blk192/*   one Block at the reference's layer_scale_init_value 1e-6 on an odd 5x7 map: output, input gradient,
           gradients of the small parameters and of the first 4 rows of the two pointwise weights
  ln/*     LayerNorm channels_last and channels_first on the same values: outputs and gradients
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_FILE = "/root/reference/semantic_segmentation/backbone/convnext.py"


def install_standins():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    timm, models, layers = mod("timm"), mod("timm.models"), mod("timm.models.layers")
    timm.models, models.layers = models, layers
    layers.trunc_normal_ = torch.nn.init.trunc_normal_

    class DropPath(torch.nn.Module):   # only reachable with drop_path > 0, which this script never passes
        def __init__(self, *a, **k):
            raise RuntimeError("DropPath stand-in must not be constructed (drop_path is 0 in every case)")

    layers.DropPath = DropPath
    mod("mmcv_custom").load_checkpoint = None
    mmseg, utils_m, models_m, builder = mod("mmseg"), mod("mmseg.utils"), mod("mmseg.models"), mod("mmseg.models.builder")
    mmseg.utils, mmseg.models, models_m.builder = utils_m, models_m, builder
    utils_m.get_root_logger = None

    class _Registry:
        def register_module(self):
            return lambda cls: cls

    builder.BACKBONES = _Registry()


def load_reference():
    install_standins()
    spec = importlib.util.spec_from_file_location("ref_convnext", REF_FILE)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def bf16_exact(t):
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_bits(t):
    return t.detach().to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def randomise(module, g, gamma_mean):
    """Recorded, non-trivial, bf16-representable parameters (so biases / LayerNorm affine / layer scale all matter)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("gamma"):
                v = gamma_mean * (1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() > 1:
                v = 0.05 * torch.randn(p.shape, generator=g)
            elif name.endswith("weight"):
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                v = 0.05 * torch.randn(p.shape, generator=g)
            p.copy_(bf16_exact(v))


def main():
    ref = load_reference()
    out = {}
    g = torch.Generator().manual_seed(20240)

    # ---- the backbone -------------------------------------------------------------------------------------------
    dims, depths = [32, 64, 96, 192], [1, 1, 1, 1]
    net = ref.ConvNeXt(in_chans=3, depths=depths, dims=dims, drop_path_rate=0.0, layer_scale_init_value=1.0, out_indices=[3])
    randomise(net, g, gamma_mean=1.0)
    x = bf16_exact(torch.randn(2, 3, 64, 64, generator=g)).requires_grad_(True)
    feats = {}
    for i in range(4):
        net.stages[i].register_forward_hook(lambda m, inp, o, i=i: feats.__setitem__(i, o))
    net.forward_features(x)
    r = torch.randn(feats[3].shape, generator=g)
    loss = (feats[3] * r).sum()
    loss.backward()
    out["net/dims"] = np.array(dims)
    out["net/x"] = bf16_bits(x)
    out["net/r"] = r.numpy()
    out["net/dx"] = x.grad.numpy()
    for i in range(4):
        out[f"net/feat{i}"] = feats[i].detach().numpy()
    for name, p in net.named_parameters():
        if name.startswith("norm"):
            continue      # the per-output norms of the segmentation backbone are not part of the classification path
        out[f"net/param/{name}"] = bf16_bits(p)
        out[f"net/grad/{name}"] = p.grad.numpy()

    # ---- single blocks at the reference's own layer-scale init -----------------------------------------------------
    for dim in (96, 192):
        blk = ref.Block(dim, drop_path=0.0, layer_scale_init_value=1e-6)
        randomise(blk, g, gamma_mean=1e-6)
        xb = bf16_exact(torch.randn(2, dim, 5, 7, generator=g)).requires_grad_(True)
        yb = blk(xb)
        rb = torch.randn(yb.shape, generator=g)
        (yb * rb).sum().backward()
        k = f"blk{dim}"
        out[f"{k}/x"], out[f"{k}/r"] = bf16_bits(xb), rb.numpy()
        out[f"{k}/y"], out[f"{k}/dx"] = yb.detach().numpy(), xb.grad.numpy()
        for name, p in blk.named_parameters():
            out[f"{k}/param/{name}"] = bf16_bits(p)
            gr = p.grad
            out[f"{k}/grad/{name}"] = (gr[:4] if name in ("pwconv1.weight", "pwconv2.weight") else gr).numpy()

    # ---- LayerNorm, both data formats ------------------------------------------------------------------------------
    C = 96
    v = bf16_exact(torch.randn(2, 5, 7, C, generator=g))
    w = bf16_exact(1.0 + 0.1 * torch.randn(C, generator=g))
    b = bf16_exact(0.05 * torch.randn(C, generator=g))
    rl = torch.randn(2, 5, 7, C, generator=g)
    out["ln/x"], out["ln/w"], out["ln/b"], out["ln/r"] = bf16_bits(v), bf16_bits(w), bf16_bits(b), rl.numpy()
    for fmt in ("channels_last", "channels_first"):
        ln = ref.LayerNorm(C, eps=1e-6, data_format=fmt)
        with torch.no_grad():
            ln.weight.copy_(w)
            ln.bias.copy_(b)
        xin = (v if fmt == "channels_last" else v.permute(0, 3, 1, 2).contiguous()).clone().requires_grad_(True)
        y = ln(xin)
        rr = rl if fmt == "channels_last" else rl.permute(0, 3, 1, 2)
        (y * rr).sum().backward()
        to_last = (lambda t: t) if fmt == "channels_last" else (lambda t: t.permute(0, 2, 3, 1).contiguous())
        out[f"ln/{fmt}/y"] = to_last(y.detach()).numpy()
        out[f"ln/{fmt}/dx"] = to_last(xin.grad).numpy()
        out[f"ln/{fmt}/dw"], out[f"ln/{fmt}/db"] = ln.weight.grad.numpy(), ln.bias.grad.numpy()

    path = os.path.join(HERE, "convnext_ref_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
