"""ConvNeXt on the HIP kernels vs the CPU oracle (oracle/convnext_ref.py, same bf16 rounding points, injected
stochastic-depth masks).  Tolerances: the oracle's own re-association noise is the yardstick (tests/test_model_gpu.py)."""
import copy

import numpy as np
import pytest
import torch

from oracle import ops_ref as R
from oracle.convnext_ref import ConvNeXtRef

pytestmark = pytest.mark.gpu


def _pair(arch, C, seed=0, drop_path=0.0):
    from imageclassification_amd.convnext import ConvNeXt
    torch.manual_seed(seed)
    ref = ConvNeXtRef(arch, C, bf16_points=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("gamma"):
                p.copy_(0.3 + 0.4 * torch.rand(p.shape, generator=g))       # layer scale far from its 1e-6 init
            elif "norm" in n or n.startswith("stem.1") or "downsample.0" in n:
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("weight"):
                p.mul_(4.0)                                                  # std 0.08: activations of O(1)
    net = ConvNeXt(arch, C, drop_path_rate=drop_path)
    net.load_state_dict(ref.state_dict())
    return ref, net


@pytest.fixture(params=[True, False], ids=["folded_layerscale", "layerscale_kernels"])
def ls_route(request, monkeypatch):
    """Both routes of the block tail: the layer scale folded into fc2 (default, ICAMD_FUSED_LAYERSCALE) and the
    icamd_layerscale_fwd / _bwd kernels of rounds 2-4."""
    import imageclassification_amd.convnext as cn
    monkeypatch.setattr(cn, "_FUSED_LS", request.param)
    return request.param


def test_convnext_forward_backward_matches_oracle(ls_route):
    from imageclassification_amd import hip
    C, B, HW = 10, 6, 64
    ref, net = _pair("convnext_test", C, drop_path=0.2)
    g = torch.Generator().manual_seed(5)
    keeps = [((torch.rand(B, generator=g) < 0.8).float() / 0.8) for _ in ref.all_blocks()]
    keeps[0] = None                                           # first block: rate 0
    for blk, k in zip(ref.all_blocks(), keeps):
        blk.keep = k
    ref64 = copy.deepcopy(ref).double()
    x = torch.randn(B, 3, HW, HW, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1)
    loss.backward()
    out64 = ref64(x.double())
    torch.nn.functional.cross_entropy(out64, y, label_smoothing=0.1).backward()
    net.train()
    net.injected_keep = [torch.ones(B) if k is None else k for k in keeps]
    ws = net.pack(x.cuda())
    logits = net.forward_packed(ws)
    yd = y.cuda()
    hip.check(net.lib.icamd_softmax_xent(ws["logits"].data_ptr(), net.ncls_p, B, C, yd.data_ptr(), None, 1.0, 0.1, 1.0 / B,
                                         ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(), ws["dlogits"].data_ptr(),
                                         hip.stream_ptr()), "xent")
    net.backward_packed(ws)
    torch.cuda.synchronize()
    got = logits[:, :C].float().cpu()
    noise = R.rel_l2(out64.detach().float(), out.detach())
    err = R.rel_l2(got, out.detach())
    assert err <= 2.0 * max(noise, 2e-3), (err, noise)
    assert abs(float(ws["loss_rows"].mean()) - float(loss)) <= 5e-3 * float(loss)
    p64 = dict(ref64.named_parameters())
    worst = ("", 0.0)
    for name, p in ref.named_parameters():
        e = R.rel_l2(net.grad_of(name), p.grad)
        n = R.rel_l2(p64[name].grad.float(), p.grad)
        if e > worst[1]:
            worst = (name, e)
        assert e <= 3.0 * max(n, 1e-2), (name, e, n)
    print(f"convnext_test: logits err {err:.2e} (self-noise {noise:.2e}); worst grad err {worst[1]:.2e} at {worst[0]}")


def test_hip_convnext_matches_reference_vectors(ls_route):
    """The HIP ConvNeXt against vectors computed by the REFERENCE's own ConvNeXt class (tests/golden/
    convnext_ref_vectors.npz <- /root/reference/semantic_segmentation/backbone/convnext.py, fp32): stage outputs and every
    parameter gradient of the 4-stage backbone (dims 32/64/96/192, layer scale ~1, non-zero biases), driven through the
    product's forward_packed / backward_packed(dfeat=...).  The reference vectors are exact fp32; the HIP path stores bf16
    activations, so the bound is bf16 noise through the depth of the net (stage outputs <= 1e-2, gradients <= 3e-2
    relative L2; a wrong tap, transposed filter or dropped bias is O(1)).  The same inputs through the oracle WITH the HIP
    path's rounding points must agree much closer (<= 2x the oracle's own fp64 re-association noise, floor 3e-3)."""
    import os
    from imageclassification_amd.convnext import ConvNeXt
    from _convnext_pin import _bf16_bits_to_f32, convnext_ref_to_timm_name
    v = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convnext_ref_vectors.npz"))
    ref = ConvNeXtRef("convnext_pin", 10, bf16_points=True)
    sd = ref.state_dict()
    names = [k[len("net/param/"):] for k in v.files if k.startswith("net/param/")]
    for n in names:
        t = convnext_ref_to_timm_name(n)
        sd[t].copy_(_bf16_bits_to_f32(v["net/param/" + n]).reshape(sd[t].shape))
    net = ConvNeXt("convnext_pin", 10)
    net.load_state_dict(ref.state_dict())
    x = _bf16_bits_to_f32(v["net/x"]).reshape(2, 3, 64, 64)
    r = torch.from_numpy(v["net/r"])                                  # [2,192,2,2] NCHW
    r16 = r.to(torch.bfloat16)
    # oracle with rounding points (+ fp64 copy for the yardstick), gradient injected at the last stage output
    ref64 = copy.deepcopy(ref).double()
    f = ref.forward_features(x)
    (f[3] * r16.float()).sum().backward()
    f64 = ref64.forward_features(x.double())
    (f64[3] * r16.double()).sum().backward()
    net.train()
    ws = net.pack(x.cuda())
    net.forward_packed(ws)
    net.backward_packed(ws, dfeat=r16.permute(0, 2, 3, 1).contiguous().cuda())
    torch.cuda.synchronize()
    worst_ref, worst_orc = ("", 0.0), ("", 0.0)
    for i in range(4):
        got = ws["stages"][i]["blocks"][-1]["out"].float().cpu().permute(0, 3, 1, 2)
        e_ref = R.rel_l2(got, torch.from_numpy(v[f"net/feat{i}"]))
        e_orc = R.rel_l2(got, f[i].detach())
        n_orc = R.rel_l2(f64[i].detach().float(), f[i].detach())
        assert e_ref <= 1e-2, (i, e_ref)
        assert e_orc <= 2.0 * max(n_orc, 3e-3), (i, e_orc, n_orc)
    params, p64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    rows = []
    for n in names:
        t = convnext_ref_to_timm_name(n)
        g = net.grad_of(t)
        e_ref = R.rel_l2(g, torch.from_numpy(v["net/grad/" + n]).reshape(g.shape))
        e_orc = R.rel_l2(g, params[t].grad)
        n_orc = R.rel_l2(p64[t].grad.float(), params[t].grad)
        worst_ref = max(worst_ref, (t, e_ref), key=lambda a: a[1])
        worst_orc = max(worst_orc, (t, e_orc), key=lambda a: a[1])
        rows.append((t, e_ref, e_orc, n_orc))
    bad = [r_ for r_ in rows if r_[1] > 3e-2 or r_[2] > 2.0 * max(r_[3], 3e-3)]
    for t, e_ref, e_orc, n_orc in (bad or []):
        print(f"  {t:40s} vs reference {e_ref:.2e}  vs oracle {e_orc:.2e} (oracle self-noise {n_orc:.2e})")
    assert not bad, [b[0] for b in bad]
    print(f"convnext_pin vs reference vectors: worst gradient rel-L2 {worst_ref[1]:.2e} at {worst_ref[0]}; vs the oracle with "
          f"bf16 rounding points {worst_orc[1]:.2e} at {worst_orc[0]}")


def test_config4_mixup_cutmix_ema_tracks_oracle_loop():
    """BASELINE configs[4]'s composition -- ConvNeXt + Mixup(0.8)/CutMix(1.0) + SoftTargetCrossEntropy + ModelEmaV3 + AdamW,
    through train_one_epoch -- against the ORACLE LOOP (oracle/engine_ref.py, pinned to the reference's engine.py) with its
    own gradients, not a finiteness check: same numpy seed on both sides, so the mixup / cutmix draws (mode, lambda, box)
    must be IDENTICAL step by step (asserted); per-step loss within 3e-3 (bf16 forward noise on slightly different weights);
    the accuracy of the un-mixed images through the updated model (engine.py:89-97) within 2 samples; the parameter and EMA
    displacement after 4 steps has the oracle's direction (cosine >= 0.9) and size; the EMA recursion
    ema <- decay*ema + (1-decay)*p holds exactly on the HIP trajectory."""
    from imageclassification_amd.ema import ModelEmaV3
    from imageclassification_amd.engine import train_one_epoch
    from imageclassification_amd.mixup import Mixup, SoftTargetCrossEntropy
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    from oracle import engine_ref as E
    C, B, HW, steps, decay = 10, 8, 64, 4, 0.9
    ref, net = _pair("convnext_test", C, seed=3)
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    data = [(torch.randn(B, 3, HW, HW, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(steps)]
    lr, wd = [1e-3] * steps, [5e-2] * steps

    class Spy(E.MixupRef):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.draws = []

        def __call__(self, x, t):
            out = super().__call__(x, t)
            self.draws.append(self.last)
            return out

    np.random.seed(7)
    mix_ref = Spy(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
    ema_ref = E.ModelEmaRef(ref, decay=decay)
    opt_ref = torch.optim.AdamW([{"params": list(ref.parameters()), "weight_decay": 5e-2}], lr=1e-3, weight_decay=0.0)
    tr = []
    E.train_one_epoch_ref(ref, E.SoftTargetCrossEntropyRef(), [(x.clone(), t.clone()) for x, t in data], opt_ref,
                          model_ema=ema_ref, mixup_fn=mix_ref, lr_schedule_values=lr, wd_schedule_values=wd,
                          num_training_steps_per_epoch=steps, num_classes=C, trace=tr)

    np.random.seed(7)
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
    draws = []
    sample = mix.sample
    mix.sample = lambda shape: draws.append(sample(shape)) or draws[-1]
    ema = ModelEmaV3(net, decay=decay)
    opt = create_optimizer("adamw", 1e-3, 5e-2, net)
    ema_expect = net.param_arena.clone()
    losses, accs = [], []
    for i in range(steps):       # one step per call: the per-step numbers and the parameters after every step are visible
        st = train_one_epoch(net, SoftTargetCrossEntropy(), [data[i]], opt, torch.device("cuda"), 0,
                             NativeScalerWithGradNormCount(), None, ema, mix, start_steps=i, lr_schedule_values=lr,
                             wd_schedule_values=wd, num_training_steps_per_epoch=1, update_freq=1, use_amp=False, num_classes=C)
        losses.append(st["loss"])
        accs.append(st["class_acc"])
        ema_expect = ema_expect + (1.0 - decay) * (net.param_arena - ema_expect)
    assert len(draws) == steps
    for (m, lam, box), (rm, rlam, rbox) in zip(draws, mix_ref.draws):
        assert m == rm and tuple(box) == tuple(rbox) and abs(lam - rlam) <= 1e-12
    assert {d[0] for d in draws} == {1, 2}            # both mixup and cutmix steps occurred
    for i in range(steps):
        assert abs(losses[i] - tr[i]["loss"]) <= 3e-3 * tr[i]["loss"], (i, losses[i], tr[i]["loss"])
        assert abs(accs[i] - tr[i]["class_acc"]) <= 2.0 / B + 1e-9
    assert torch.allclose(ema.param_arena, ema_expect, rtol=1e-5, atol=1e-7)
    sd, esd, rsd, resd = net.state_dict(), ema.module.state_dict(), ref.state_dict(), ema_ref.module.state_dict()
    for what, got, want in (("params", sd, rsd), ("ema", esd, resd)):
        dg = torch.cat([(got[k] - init[k]).flatten() for k in init])
        dw = torch.cat([(want[k] - init[k]).flatten() for k in init])
        cos = float((dg * dw).sum() / (dg.norm() * dw.norm()))
        print(f"config4 composition: {what} displacement cosine {cos:.4f}, size ratio {float(dg.norm() / dw.norm()):.4f}")
        assert cos >= 0.9 and 0.9 <= float(dg.norm() / dw.norm()) <= 1.1, (what, cos)


def test_convnext_tiny_config4_composition():
    """BASELINE configs[4]: ConvNeXt-T + mixup/cutmix + model EMA through the engine (small batch, 2 steps)."""
    from imageclassification_amd.convnext import ConvNeXt
    from imageclassification_amd.ema import ModelEmaV3
    from imageclassification_amd.engine import evaluate, train_one_epoch
    from imageclassification_amd.mixup import Mixup, SoftTargetCrossEntropy
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    C, B = 1000, 4
    net = ConvNeXt("convnext_tiny", C, drop_path_rate=0.05, seed=3)
    assert sum(int(np.prod(p.torch_shape)) for p in net.params.values()) == 28589128
    sd = net.state_dict()
    assert sd["stages.1.downsample.1.weight"].shape == (192, 96, 2, 2) and sd["stages.0.blocks.0.conv_dw.weight"].shape == (96, 1, 7, 7)
    ema = ModelEmaV3(net, decay=0.9995)
    opt = create_optimizer("adamw", 1e-3, 5e-2, net)
    np.random.seed(0)
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
    g = torch.Generator().manual_seed(1)
    data = [(torch.randn(B, 3, 224, 224, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(2)]
    stats = train_one_epoch(net, SoftTargetCrossEntropy(), data, opt, torch.device("cuda"), 0, NativeScalerWithGradNormCount(),
                            None, ema, mix, start_steps=0, lr_schedule_values=[1e-4, 2e-4], wd_schedule_values=[5e-2, 5e-2],
                            num_training_steps_per_epoch=2, update_freq=1, use_amp=True, num_classes=C)
    assert np.isfinite(stats["loss"]) and 6.0 < stats["loss"] < 8.0        # ~ln(1000) at init
    assert opt.step_count == 2
    ev = evaluate(data, ema.module, torch.device("cuda"), C)
    assert np.isfinite(ev["loss"]) and "acc1" in ev
