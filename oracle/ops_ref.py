"""CPU oracle for the per-step device operations (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Each function restates, with torch-CPU fp32 primitives, the arithmetic one C-ABI entry point of
include/icamd.h performs, with the SAME bf16 rounding points as the HIP path (inputs are bf16 values held in
fp32, accumulation is fp32, the result is rounded once).  The reference itself has no kernels: the arithmetic
it relies on is what torch runs for timm's layers under /root/reference/engine.py:48,51,64,72 and for
torch.optim.AdamW / timm ModelEmaV3 / timm Mixup (engine.py:44,68,74,77); torch-CPU is therefore the pin.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import torch
import torch.nn.functional as F


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


# ---- convolution: icamd_conv2d_fwd / _dgrad / _wgrad -----------------------------------------------
def conv2d_fwd(x_nhwc, w_krsc, stride, pad, bias=None, addend=None):
    """x [N,H,W,Cin], w [Cout,KH,KW,Cin] (bf16-representable fp32). Returns (y rounded to bf16 grid, fp32)."""
    x = nhwc_to_nchw(x_nhwc.float())
    w = w_krsc.float().permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(x, w, None, stride=stride, padding=pad)
    y = nchw_to_nhwc(y)
    if bias is not None:
        y = y + bias.float()
    if addend is not None:
        y = y + addend.float()
    return bf16_round(y)


def conv2d_stats(y_nhwc):
    """Per-channel sum and sum of squares of the rounded output (fp64 accumulate)."""
    y = y_nhwc.double().reshape(-1, y_nhwc.shape[-1])
    return y.sum(0), (y * y).sum(0)


def conv2d_dgrad(dy_nhwc, w_krsc, in_hw, stride, pad, addend=None):
    dy = nhwc_to_nchw(dy_nhwc.float())
    w = w_krsc.float().permute(0, 3, 1, 2).contiguous()
    N = dy.shape[0]
    dx = torch.nn.grad.conv2d_input((N, w.shape[1], in_hw[0], in_hw[1]), w, dy, stride=stride, padding=pad)
    dx = nchw_to_nhwc(dx)
    if addend is not None:
        dx = dx + addend.float()
    return bf16_round(dx)


def conv2d_wgrad(x_nhwc, dy_nhwc, ksize, stride, pad):
    """Returns dw [Cout,KH,KW,Cin] fp32 (not rounded: the HIP path keeps weight gradients in fp32)."""
    x = nhwc_to_nchw(x_nhwc.float())
    dy = nhwc_to_nchw(dy_nhwc.float())
    Cout, Cin = dy.shape[1], x.shape[1]
    dw = torch.nn.grad.conv2d_weight(x, (Cout, Cin, ksize[0], ksize[1]), dy, stride=stride, padding=pad)
    return dw.permute(0, 2, 3, 1).contiguous()


# ---- BatchNorm: icamd_bn_train_finalize / _apply / _bwd ---------------------------------------------
def bn_train_coeffs(y_nhwc, gamma, beta, running_mean, running_var, momentum, eps):
    """Batch statistics of the (bf16-rounded) conv output; returns mean, invstd, scale, shift, new running stats."""
    C = y_nhwc.shape[-1]
    y = y_nhwc.double().reshape(-1, C)
    n = y.shape[0]
    mean = y.mean(0)
    var = (y * y).mean(0) - mean * mean
    var = var.clamp_min(0)
    invstd = (1.0 / torch.sqrt(var + eps)).float()
    meanf = mean.float()
    scale = gamma.float() * invstd
    shift = beta.float() - meanf * scale
    unbiased = var * (n / (n - 1)) if n > 1 else var
    rm = (1 - momentum) * running_mean.float() + momentum * meanf
    rv = (1 - momentum) * running_var.float() + momentum * unbiased.float()
    return meanf, invstd, scale, shift, rm, rv


def bn_apply(y_nhwc, scale, shift, residual=None, relu=True):
    out = torch.addcmul(shift.float(), y_nhwc.float(), scale.float())  # fma(y, scale, shift)
    if residual is not None:
        out = out + residual.float()
    if relu:
        out = out.clamp_min(0)
    return bf16_round(out)


def bn_bwd(dout, act, y, mean, invstd, scale, relu=True):
    """dout, act (post-activation output, or None -> recompute the mask), y: NHWC. Returns dy, dgamma, dbeta, g."""
    C = y.shape[-1]
    g = dout.float()
    if relu:
        if act is not None:
            mask = act.float() > 0
        else:
            raise ValueError("oracle needs the activation for the mask")
        g = g * mask
    xhat = (y.float() - mean.float()) * invstd.float()
    g2 = g.reshape(-1, C)
    xh2 = xhat.reshape(-1, C)
    n = g2.shape[0]
    sg = g2.double().sum(0)
    sgx = (g2.double() * xh2.double()).sum(0)
    c1 = (sg / n).float()
    c2 = (sgx / n).float()
    dy = scale.float() * (g - c1 - xhat * c2)
    return bf16_round(dy), sgx.float(), sg.float(), bf16_round(g)


# ---- pooling ---------------------------------------------------------------------------------------
def maxpool3x3s2_fwd(x_nhwc):
    x = nhwc_to_nchw(x_nhwc.float())
    out, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
    return nchw_to_nhwc(out), idx


def maxpool3x3s2_bwd(dout_nhwc, x_nhwc):
    x = nhwc_to_nchw(x_nhwc.float()).requires_grad_(True)
    out = F.max_pool2d(x, 3, 2, 1)
    out.backward(nhwc_to_nchw(dout_nhwc.float()))
    return bf16_round(nchw_to_nhwc(x.grad))


def avgpool_fwd(x_nhwc):
    return bf16_round(x_nhwc.float().mean(dim=(1, 2)))


def avgpool_bwd(dout_nc, hw):
    N, C = dout_nc.shape
    return bf16_round((dout_nc.float() / hw).reshape(N, 1, C).expand(N, hw, C).contiguous())


# ---- input packing + mixup / cutmix (timm.data.Mixup batch mode) ------------------------------------
def pack_input(x_nchw, mode=0, lam=1.0, box=None):
    x = x_nchw.float().clone()
    if mode == 1:
        x = x * lam + x.flip(0) * (1.0 - lam)
    elif mode == 2:
        yl, yh, xl, xh = box
        x[:, :, yl:yh, xl:xh] = x_nchw.float().flip(0)[:, :, yl:yh, xl:xh]
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, 8)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return bf16_round(out)


# ---- loss ------------------------------------------------------------------------------------------
def soft_targets(y1, y2, lam, smoothing, C):
    off = smoothing / C
    on = 1.0 - smoothing + off
    t1 = torch.full((y1.shape[0], C), off).scatter_(1, y1.view(-1, 1), on)
    t2 = torch.full((y2.shape[0], C), off).scatter_(1, y2.view(-1, 1), on)
    return t1 * lam + t2 * (1.0 - lam)


def softmax_xent(logits, y1, y2=None, lam=1.0, smoothing=0.0, gscale=1.0):
    """logits [B,C] (bf16-representable). Returns per-row loss, argmax, dlogits (bf16 grid)."""
    x = logits.float().requires_grad_(True)
    B, C = x.shape
    t = soft_targets(y1, y1 if y2 is None else y2, lam, smoothing, C)
    logp = F.log_softmax(x, dim=-1)
    loss_rows = -(t * logp).sum(-1)
    (loss_rows.sum() * gscale).backward()
    return loss_rows.detach(), x.detach().argmax(-1), bf16_round(x.grad)


# ---- optimizer -------------------------------------------------------------------------------------
def adamw_ema_steps(p0, grads, lrs, wds, betas=(0.9, 0.999), eps=1e-8, ema0=None, ema_decay=0.9995, gscale=1.0):
    """Runs torch.optim.AdamW on a flat parameter for len(grads) steps with per-step lr/wd injection
    (reference: /root/reference/engine.py:33-38) and the ModelEmaV3 lerp after every step."""
    p = torch.nn.Parameter(p0.clone().float())
    opt = torch.optim.AdamW([{"params": [p], "weight_decay": wds[0]}], lr=lrs[0], betas=betas, eps=eps, weight_decay=0.0)
    ema = None if ema0 is None else ema0.clone().float()
    for g, lr, wd in zip(grads, lrs, wds):
        for group in opt.param_groups:
            group["lr"] = lr
            if group["weight_decay"] > 0:
                group["weight_decay"] = wd
        p.grad = (g.float() * gscale).clone()
        opt.step()
        opt.zero_grad()
        if ema is not None:
            ema.lerp_(p.detach(), 1.0 - ema_decay)
    st = opt.state[p]
    return p.detach(), st["exp_avg"], st["exp_avg_sq"], ema


class LionRef(torch.optim.Optimizer):
    """timm.optim.Lion restated (timm is absent here; algorithm of Chen et al. 2023 as timm implements it and as the
    reference constructs it, /root/reference/optim_factory.py:76-77: betas=(0.9, 0.999), decoupled weight decay):
        p *= 1 - lr*wd ; p -= lr * sign(beta1*m + (1-beta1)*g) ; m = beta2*m + (1-beta2)*g.     Parity unpinned."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                m = st["exp_avg"]
                p.mul_(1.0 - group["lr"] * group["weight_decay"])
                p.add_(torch.sign(m * b1 + p.grad * (1.0 - b1)), alpha=-group["lr"])
                m.lerp_(p.grad, 1.0 - b2)


def optimizer_ema_steps(name, p0, grads, lrs, wds, ema0=None, ema_decay=0.9995, gscale=1.0):
    """The reference's non-default optimizers (optim_factory.py:66-77) for len(grads) steps with the per-step lr / wd
    injection of engine.py:33-38 and the ModelEmaV3 lerp: name in sgd|nesterov|momentum|adam|lion.
    Returns (p, first-moment / momentum buffer, second moment or None, ema)."""
    p = torch.nn.Parameter(p0.clone().float())
    groups = [{"params": [p], "weight_decay": wds[0]}]
    if name in ("sgd", "nesterov"):
        opt = torch.optim.SGD(groups, lr=lrs[0], momentum=0.9, nesterov=True, weight_decay=0.0)
    elif name == "momentum":
        opt = torch.optim.SGD(groups, lr=lrs[0], momentum=0.9, nesterov=False, weight_decay=0.0)
    elif name == "adam":
        opt = torch.optim.Adam(groups, lr=lrs[0], weight_decay=0.0)
    elif name == "lion":
        opt = LionRef(groups, betas=(0.9, 0.999))
    else:
        raise ValueError(name)
    ema = None if ema0 is None else ema0.clone().float()
    for g, lr, wd in zip(grads, lrs, wds):
        for group in opt.param_groups:
            group["lr"] = lr
            if group["weight_decay"] > 0:
                group["weight_decay"] = wd
        p.grad = (g.float() * gscale).clone()
        opt.step()
        opt.zero_grad()
        if ema is not None:
            ema.lerp_(p.detach(), 1.0 - ema_decay)
    st = opt.state[p]
    m = st.get("momentum_buffer", st.get("exp_avg"))
    return p.detach(), m, st.get("exp_avg_sq"), ema


def grad_norm(g, max_norm=0.0):
    norm = torch.linalg.vector_norm(g.double()).float()
    coef = 1.0
    if max_norm > 0:
        coef = min(1.0, max_norm / (float(norm) + 1e-6))
    return float(norm), coef


# ---- comparison helpers ----------------------------------------------------------------------------
def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    d = torch.linalg.vector_norm(a - b)
    n = torch.linalg.vector_norm(b)
    return float(d / n) if n > 0 else float(d)


def max_bf16_ulp(a, b):
    """Largest |a-b| in units of the bf16 spacing at |b| (both already on the bf16 grid)."""
    a = a.float().flatten()
    b = b.float().flatten()
    mag = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    return float(((a - b).abs() / ulp).max())


def bf16_close(a, b, ulps=2.0, atol_rms=2e-3, max_frac=0.0):
    """Elementwise: |a-b| <= ulps * bf16 spacing at |b| + atol_rms * rms(b).
    The absolute term covers outputs that are small differences of O(rms) terms (cancellation), where a
    different fp32 summation order legitimately moves the result by many ulps OF THE RESULT.
    max_frac: share of the elements that may sit outside the bound (tests over millions of elements whose operands pass
    through a bf16 rounding: a handful land one rounding step further out)."""
    a = a.float().flatten()
    b = b.float().flatten()
    mag = torch.maximum(a.abs(), b.abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    rms = float(torch.sqrt((b.double() ** 2).mean()))
    # "not within the bound" (rather than "beyond it"): a NaN / Inf in either tensor compares False and counts as bad
    # (an Inf makes its own bound infinite, hence the explicit finiteness term)
    bad = ~((a - b).abs() <= ulps * ulp + atol_rms * rms) | ~torch.isfinite(a) | ~torch.isfinite(b)
    if max_frac <= 0.0:
        return not bool(bad.any())
    return bool(float(bad.float().mean()) <= max_frac)


# ---- token ops: LayerNorm / GELU / attention (ViT, ConvNeXt) ---------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps):
    """x [rows, C] bf16-representable. Returns y (bf16 grid), mean, rstd."""
    xf = x.float()
    mean = xf.mean(-1)
    var = ((xf - mean[:, None]) ** 2).mean(-1)
    rstd = torch.rsqrt(var + eps)
    y = (xf - mean[:, None]) * rstd[:, None] * gamma.float() + beta.float()
    return bf16_round(y), mean, rstd


def layernorm_bwd(dy, x, gamma, eps):
    """Gradients through torch's own layer_norm (fp32): dx (bf16 grid), dgamma, dbeta."""
    xf = x.float().requires_grad_(True)
    g = gamma.float().clone().requires_grad_(True)
    b = torch.zeros_like(g).requires_grad_(True)
    y = F.layer_norm(xf, (x.shape[-1],), g, b, eps)
    y.backward(dy.float())
    return bf16_round(xf.grad), g.grad, b.grad


def gelu_fwd(z):
    return bf16_round(F.gelu(z.float()))


def gelu_bwd(da, z):
    zf = z.float().requires_grad_(True)
    F.gelu(zf).backward(da.float())
    return bf16_round(zf.grad)


def attention_fwd(qkv, B, T, H, D, scale):
    """qkv [B*T, 3*H*D] (timm layout: reshape(B, T, 3, H, D)). Returns out [B*T, H*D] (bf16 grid), lse [B,H,T]."""
    q, k, v = qkv.float().reshape(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, dim=-1)
    p = torch.softmax(s, dim=-1)
    o = p @ v
    return bf16_round(o.permute(0, 2, 1, 3).reshape(B * T, H * D)), lse


def attention_bwd(qkv, dout, B, T, H, D, scale):
    x = qkv.float().requires_grad_(True)
    q, k, v = x.reshape(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    o = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v
    o.permute(0, 2, 1, 3).reshape(B * T, H * D).backward(dout.float())
    return bf16_round(x.grad)


# ---- ConvNeXt pieces ------------------------------------------------------------------------------------------------
def dwconv7_fwd(x_nhwc, w_c77, bias):
    """x [N,H,W,C], w [C,7,7] (torch depthwise layout), bias [C]. Returns y NHWC on the bf16 grid."""
    C = x_nhwc.shape[-1]
    y = F.conv2d(nhwc_to_nchw(x_nhwc.float()), w_c77.float().reshape(C, 1, 7, 7), bias.float(), padding=3, groups=C)
    return bf16_round(nchw_to_nhwc(y))


def dwconv7_bwd(x_nhwc, w_c77, dy_nhwc, addend=None):
    C = x_nhwc.shape[-1]
    x = nhwc_to_nchw(x_nhwc.float()).requires_grad_(True)
    w = w_c77.float().reshape(C, 1, 7, 7).clone().requires_grad_(True)
    F.conv2d(x, w, None, padding=3, groups=C).backward(nhwc_to_nchw(dy_nhwc.float()))
    dx = nchw_to_nhwc(x.grad)
    if addend is not None:
        dx = dx + addend.float()
    return bf16_round(dx), w.grad.reshape(C, 7, 7)


def layerscale_fwd(z, inp, gamma, keep):
    k = 1.0 if keep is None else keep.float().reshape(-1, *([1] * (z.dim() - 1)))
    return bf16_round(inp.float() + k * gamma.float() * z.float())


def layerscale_bwd(dout, z, gamma, keep):
    k = 1.0 if keep is None else keep.float().reshape(-1, *([1] * (z.dim() - 1)))
    d = dout.float() * k
    return bf16_round(d * gamma.float()), (d * z.float()).reshape(-1, z.shape[-1]).double().sum(0).float()
