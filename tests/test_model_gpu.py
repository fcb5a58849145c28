"""End-to-end parity of the HIP ResNet (forward, loss, backward, AdamW+EMA) against the CPU oracle network
with the same bf16 rounding points (oracle/resnet_ref.py), on seeded inputs small enough for CPU seconds.

Tolerances are written at each check.  Per-kernel parity is at the north_star's 1e-3 (tests/test_kernels_gpu.py);
whole-network comparisons are calibrated against the oracle's own re-association noise, and the backward
wiring is proven with a teacher-forced run (both explained at the tests).
"""
import ctypes

import pytest
import torch

from oracle import ops_ref as R
from oracle.resnet_ref import ResNetRef

pytestmark = pytest.mark.gpu


def _pair(arch, num_classes, seed=0):
    from imageclassification_amd.nets import ResNet
    torch.manual_seed(seed)
    ref = ResNetRef(arch, num_classes, bf16_points=True, zero_init_last=False)
    # non-trivial BN affine parameters so their gradients are exercised
    g = torch.Generator().manual_seed(seed + 1)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.1 * torch.randn(m.bias.shape, generator=g)
    net = ResNet(arch, num_classes)
    net.load_state_dict(ref.state_dict())
    return ref, net


def _xent_backward(net, ws, targets, num_classes, smoothing=0.0):
    from imageclassification_amd import hip
    lib = net.lib
    B = targets.shape[0]
    hip.check(lib.icamd_softmax_xent(ws["logits"].data_ptr(), net.ncls_p, B, num_classes, targets.data_ptr(), None, 1.0,
                                     smoothing, 1.0 / B, ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(),
                                     ws["dlogits"].data_ptr(), hip.stream_ptr()), "xent")
    net.backward_packed(ws)
    torch.cuda.synchronize()
    return float(ws["loss_rows"].mean())


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def test_resnet18_forward_backward_within_asserted_reassociation_noise():
    """ResNet-18 at timm's default init, batch 16 at 96x96.  As for ResNet-50 below: the yardstick (the oracle against its
    own fp64-accumulating copy) is asserted small first, then the HIP path is held to 2x of it.  (Round 1's version of this
    test used random BatchNorm weights on every layer at batch 4-8 / 64x64, where the ORACLE's own gradient noise is 0.15-1.0
    -- ReLU-mask flips cost sqrt(fraction flipped) in relative L2 -- and so accepted anything; that configuration is now
    only used teacher-forced, where the masks are pinned.)"""
    import copy
    C, B, HW = 10, 16, 96
    ref, net = _timm_default_pair("resnet18", C)
    ref64 = copy.deepcopy(ref).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, HW, HW, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    ref.train(); ref64.train()
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1)
    loss.backward()
    out64 = ref64(x.double())
    torch.nn.functional.cross_entropy(out64, y, label_smoothing=0.1).backward()
    net.train()
    ws = net.pack(x.cuda())
    logits = net.forward_packed(ws)
    hip_loss = _xent_backward(net, ws, y.cuda(), C, smoothing=0.1)
    got = logits[:, :C].float().cpu()
    noise_logits = R.rel_l2(out64.detach().float(), out.detach())
    p64 = dict(ref64.named_parameters())
    rows = []
    for name, p in ref.named_parameters():
        if float(p.grad.abs().max()) == 0.0:
            assert float(net.grad_of(name).abs().max()) == 0.0, name
            continue
        rows.append((name, R.rel_l2(net.grad_of(name), p.grad), R.rel_l2(p64[name].grad.float(), p.grad)))
    mean_e = sum(r[1] for r in rows) / len(rows)
    mean_n = sum(r[2] for r in rows) / len(rows)
    print(f"resnet18: logits err {R.rel_l2(got, out.detach()):.2e} (self-noise {noise_logits:.2e}); mean grad err "
          f"{mean_e:.2e} (self-noise {mean_n:.2e}) over {len(rows)} tensors")
    assert noise_logits <= 5e-3 and mean_n <= 2e-2, (noise_logits, mean_n)
    assert R.rel_l2(got, out.detach()) <= 2.0 * max(noise_logits, 1e-3)
    assert abs(hip_loss - float(loss)) <= 1e-3 * abs(float(loss))
    assert mean_e <= 2.0 * max(mean_n, 1e-3)
    for name, e, n in rows:
        assert e <= 3.0 * max(n, 5e-3), (name, e, n)


def test_resnet18_odd_input_size_matches_oracle():
    """Input sizes the stem layout does not fit exactly (odd width, odd height): the packed image gets one more zero column
    and the result is that of the true size -- train-mode loss / logits and the stem filter gradient against the oracle,
    eval-mode (BatchNorm-folded) logits too."""
    C, B = 10, 8
    ref, net = _timm_default_pair("resnet18", C, seed=4)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 3, 67, 61, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    ref.train()
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    net.train()
    ws = net.pack(x.cuda())
    logits = net.forward_packed(ws)
    hip_loss = _xent_backward(net, ws, y.cuda(), C)
    assert R.rel_l2(logits[:, :C].float().cpu(), out.detach()) <= 1e-2
    assert abs(hip_loss - float(loss)) <= 5e-3 * abs(float(loss))
    assert R.rel_l2(net.grad_of("conv1.weight"), ref.conv1.weight.grad) <= 8e-2      # whole backward chain in bf16
    assert R.rel_l2(net.grad_of("fc.weight"), ref.fc.weight.grad) <= 2e-2
    ref.eval(); net.eval()
    net.load_state_dict(ref.state_dict())          # same running statistics on both sides
    with torch.no_grad():
        eo = ref(x)
    assert R.rel_l2(net(x.cuda()).float().cpu(), eo) <= 1e-2


def _timm_default_pair(arch, num_classes, seed=0):
    """The configuration the reference really trains from: timm's ResNet init (Kaiming fan-out filters, BatchNorm weight 1 /
    bias 0, ZERO-initialised last BatchNorm weight of every residual block; /root/reference/train.py:194 create_model)."""
    from imageclassification_amd.nets import ResNet
    torch.manual_seed(seed)
    ref = ResNetRef(arch, num_classes, bf16_points=True, zero_init_last=True)
    net = ResNet(arch, num_classes)
    net.load_state_dict(ref.state_dict())
    return ref, net


class _As64(torch.nn.Module):
    """fp64 copy of the oracle network fed with the same fp32 batches (the re-association yardstick)."""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x):
        return self.m(x.double())


@pytest.mark.parametrize("gamma_last", [0.0, 0.02])
def test_resnet50_whole_network_parity_well_conditioned(gamma_last):
    """ResNet-50 (the headline network), batch 32 at 128x128: every BatchNorm reduces over >= 512 values.  The yardstick --
    the oracle against its own fp64-accumulating copy, same bf16 rounding points -- is ASSERTED to be small first, so this
    test cannot degenerate into accepting anything (an all-zero or mis-wired gradient scores 1.0, a sign error 2.0); the HIP
    path must then sit within 2x of it (floors: north_star's 1e-3 on logits / loss).

    gamma_last = 0: timm's default init, the configuration the reference really trains from.  With the last BatchNorm weight
    of each block at zero the residual branches receive an exactly-zero gradient in both implementations (asserted EXACTLY
    zero on the HIP side): 49 of 161 tensors are compared, yardstick logits <= 5e-3, gradients mean <= 4e-2 / worst <= 6e-2.

    gamma_last = 0.02: the same network with every block's last BatchNorm weight at 0.02, so that ALL 161 gradient tensors
    -- every bottleneck conv1 / conv2 / conv3, bn1 / bn2: every 3x3 data and weight gradient -- are non-zero and compared at
    t = 0.  Why 0.02 and not 0.25-0.5: bf16 training differs between any two correct implementations by ReLU-mask flips
    (a fraction f of flipped bits costs sqrt(f) in relative L2 of a gradient), and residual branches with O(1) weight
    amplify forward differences block after block: measured on the CPU oracle alone, fp64 vs fp32 accumulation, the
    gradients differ by a MEAN relative L2 of 0.30 at gamma 0.25 and 0.59 at gamma 0.5 (worst 0.43 / 0.79) -- a yardstick
    that cannot tell a wrong gradient from a right one.  At 0.02 the branches do not amplify, the gradient only crosses their
    two inner ReLU masks: yardstick mean 8.7e-2, worst 1.5e-1 (a BatchNorm bias in layer4: a sum with cancellation), asserted
    <= 0.13 / 0.22; a dropped term, tap or stride class in any branch kernel scores >= 0.5 on its tensor."""
    import copy
    C, B, HW = 100, 32, 128
    ref, net = _timm_default_pair("resnet50", C)
    if gamma_last:
        for n, m in ref.named_modules():
            if n.endswith("bn3"):
                m.weight.data.fill_(gamma_last)
        net.load_state_dict(ref.state_dict())
    ref64 = copy.deepcopy(ref).double()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, HW, HW, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    ref.train(); ref64.train()
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1)
    loss.backward()
    out64 = ref64(x.double())
    loss64 = torch.nn.functional.cross_entropy(out64, y, label_smoothing=0.1)
    loss64.backward()

    net.train()
    ws = net.pack(x.cuda())
    logits = net.forward_packed(ws)
    hip_loss = _xent_backward(net, ws, y.cuda(), C, smoothing=0.1)
    got = logits[:, :C].float().cpu()
    noise_logits = R.rel_l2(out64.detach().float(), out.detach())
    err_logits = R.rel_l2(got, out.detach())
    p64 = dict(ref64.named_parameters())
    rows, zero_branch = [], 0
    for name, p in ref.named_parameters():
        if float(p.grad.abs().max()) == 0.0:
            assert float(net.grad_of(name).abs().max()) == 0.0, name     # zero-gamma branches: exactly zero on both sides
            zero_branch += 1
            continue
        rows.append((name, R.rel_l2(net.grad_of(name), p.grad), R.rel_l2(p64[name].grad.float(), p.grad)))
    mean_e = sum(r[1] for r in rows) / len(rows)
    mean_n = sum(r[2] for r in rows) / len(rows)
    worst = max(rows, key=lambda r: r[1])
    print(f"resnet50 B={B} {HW}x{HW} gamma_last={gamma_last}: logits err {err_logits:.2e} (self-noise {noise_logits:.2e}); "
          f"loss {hip_loss:.6f} vs {float(loss):.6f}; {len(rows)} gradient tensors: mean err {mean_e:.2e} (self-noise "
          f"{mean_n:.2e}), worst {worst[0]} {worst[1]:.2e} (its self-noise {worst[2]:.2e}); {zero_branch} zero-gradient "
          f"branch tensors exact")
    if gamma_last:
        for kind in ("conv1.weight", "conv2.weight", "conv3.weight", "bn1.weight", "bn2.weight", "bn3.weight"):
            sel = [r for r in rows if r[0].endswith(kind)]
            print(f"  {kind:14s} {len(sel):3d} tensors: HIP mean {sum(r[1] for r in sel) / len(sel):.2e} worst "
                  f"{max(r[1] for r in sel):.2e} | yardstick mean {sum(r[2] for r in sel) / len(sel):.2e} worst "
                  f"{max(r[2] for r in sel):.2e}")
    # the yardstick itself
    if gamma_last:
        assert zero_branch == 0 and len(rows) == 161            # every parameter tensor has a non-zero gradient
        assert noise_logits <= 8e-3 and mean_n <= 0.13 and max(r[2] for r in rows) <= 0.22, (noise_logits, mean_n)
    else:
        assert noise_logits <= 5e-3 and mean_n <= 4e-2 and max(r[2] for r in rows) <= 6e-2, (noise_logits, mean_n)
        assert zero_branch >= 48 * 2 and len(rows) >= 40
    # the HIP path against it
    assert err_logits <= 2.0 * max(noise_logits, 1e-3)
    assert abs(hip_loss - float(loss)) <= 1e-3 * abs(float(loss))
    assert mean_e <= 2.0 * max(mean_n, 1e-3)
    for name, e, n in rows:
        assert e <= 3.0 * max(n, 5e-3), (name, e, n)


@pytest.mark.parametrize("arch,B,HW,tol", [("resnet18", 8, 64, 3e-2), ("resnet50", 4, 96, 8e-2)])
def test_backward_teacher_forced_is_tight(arch, B, HW, tol):
    """Wiring proof without the chaos: every tensor the backward pass reads (conv outputs, activations, pooling
    indices, BatchNorm statistics, logits) is overwritten with the ORACLE's forward values, so ReLU masks agree and
    the backward is a deterministic function of identical inputs.  What remains is bf16 re-rounding of the
    activation-gradient chain (no mask flips), growing smoothly with distance from the loss: per-tensor relative
    L2 <= 3e-2 at the far end of ResNet-18's chain, <= 8e-2 for ResNet-50's 3x longer one (printed profile);
    a dropped residual / shortcut / stride class would show as O(1) from that block on."""
    from imageclassification_amd import hip
    C = 10
    ref, net = _pair(arch, C, seed=2)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 3, HW, HW, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    tr = {}
    ref.set_trace(tr)
    ref.train()
    out = ref(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()

    net.train()
    ws = net.pack(x.cuda())
    net.forward_packed(ws)
    torch.cuda.synchronize()

    def put_stats(bn, y_nchw):
        yv = y_nchw.permute(0, 2, 3, 1).float()
        sd = ref.state_dict()
        mean, invstd, scale, shift, _, _ = R.bn_train_coeffs(yv, sd[bn.name + ".weight"], sd[bn.name + ".bias"],
                                                             torch.zeros(bn.c), torch.ones(bn.c), 0.1, 1e-5)
        st = torch.cat([mean, invstd, scale, shift]).cuda()
        net.stat_arena[bn.stat_offset:bn.stat_offset + 4 * bn.c].copy_(st)

    ws["y0"].copy_(_nhwc(tr["y0"]))
    ws["a0"].copy_(_nhwc(tr["a0"]))
    put_stats(net.stem_bn, tr["y0"])
    d0 = net.stem_conv.desc(B, HW, HW)
    hip.check(net.lib.icamd_maxpool3x3s2_fwd(ws["a0"].data_ptr(), ws["p0"].data_ptr(), ws["p0_idx"].data_ptr(), B, d0.OH,
                                             d0.OW, 64, hip.stream_ptr()), "maxpool")
    for blk, b in zip(net.blocks, ws["blocks"]):
        n = blk["name"]
        for i, bn in enumerate(blk["bns"]):
            b["y"][i].copy_(_nhwc(tr[f"{n}.{i}.y"]))
            b["a"][i].copy_(_nhwc(tr[f"{n}.{i}.a"]))
            put_stats(bn, tr[f"{n}.{i}.y"])
        # the backward reads the block-output ReLU mask as 1 bit per element: rebuild it from the oracle activation
        last = _nhwc(tr[f"{n}.{len(blk['bns']) - 1}.a"]).float() > 0
        packed = (last.reshape(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8)
        b["mask"].copy_(packed.cuda())
        if "down_conv" in blk:
            b["yd"].copy_(_nhwc(tr[f"{n}.down.y"]))
            b["ad"].copy_(_nhwc(tr[f"{n}.down.a"]))
            put_stats(blk["down_bn"], tr[f"{n}.down.y"])
    ws["pooled"].copy_(tr["pooled"].to(torch.bfloat16))
    ws["logits"][:, :C].copy_(tr["logits"].to(torch.bfloat16))
    hip_loss = _xent_backward(net, ws, y.cuda(), C)
    assert abs(hip_loss - float(loss)) <= 1e-4 * abs(float(loss)) + 1e-5
    worst = ("", 0.0)
    for name, p in ref.named_parameters():
        e = R.rel_l2(net.grad_of(name), p.grad)
        if e > worst[1]:
            worst = (name, e)
        if name.endswith("conv1.weight"):
            print(f"  {name:32s} {e:.2e}")
        assert e <= tol, (name, e)
    assert R.rel_l2(net.grad_of("fc.weight"), ref.fc.weight.grad) <= 1e-3
    print(f"{arch}: teacher-forced worst grad rel-L2 = {worst[1]:.2e} at {worst[0]}")


def test_eval_mode_and_call_interface():
    C = 10
    ref, net = _pair("resnet18", C, seed=3)
    x = torch.randn(6, 3, 64, 64, generator=torch.Generator().manual_seed(9))
    ref.eval()
    net.eval()
    with torch.no_grad():
        out = ref(x)
    got = net(x.cuda()).float().cpu()
    assert got.shape == (6, C)
    # The default eval forward is the BatchNorm-folded one (different bf16 rounding points from the training path the
    # bf16_points oracle mirrors): both are priced against the oracle WITHOUT rounding points, i.e. the reference's fp32
    # arithmetic, and the unfolded path additionally against its own mirror.
    exact = ResNetRef("resnet18", C, bf16_points=False, zero_init_last=False)
    exact.load_state_dict(ref.state_dict())
    exact.eval()
    with torch.no_grad():
        truth = exact(x)
    net.fold_eval = False
    plain = net(x.cuda()).float().cpu()
    net.fold_eval = True
    assert R.rel_l2(plain, out) <= 5e-3
    e_fold, e_plain = R.rel_l2(got, truth), R.rel_l2(plain, truth)
    print(f"eval logits vs fp32 reference: folded {e_fold:.2e}, separate BatchNorm pass {e_plain:.2e}")
    assert e_fold <= 1e-2 and e_fold <= 1.5 * e_plain
    assert net.num_batches_tracked == 0


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_eval_fast_path_folds_batchnorm(arch):
    """eval forward with BatchNorm folded into the filters (one kernel per convolution) against the oracle's
    eval forward and against the unfolded HIP eval path, with non-trivial running statistics and affine terms; the fold is
    redone after the weights change."""
    C = 10
    ref, net = _pair(arch, C, seed=4)
    g = torch.Generator().manual_seed(10)
    sd = ref.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.2
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith("bn3.weight") or k.endswith("downsample.1.weight"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        elif "bn" in k and k.endswith(".bias"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
    ref.load_state_dict(sd)
    net.load_state_dict(sd)
    x = torch.randn(4, 3, 64, 64, generator=g)
    exact = ResNetRef(arch, C, bf16_points=False, zero_init_last=False)   # the reference's fp32 arithmetic
    exact.load_state_dict(sd)
    exact.eval()
    net.eval()
    with torch.no_grad():
        want = exact(x)
    assert net.fold_eval
    folded = net(x.cuda()).float().cpu()
    net.fold_eval = False
    unfolded = net(x.cuda()).float().cpu()
    net.fold_eval = True
    e_fold, e_plain = R.rel_l2(folded, want), R.rel_l2(unfolded, want)
    print(f"{arch} eval logits vs fp32 reference: folded {e_fold:.2e}, separate BatchNorm pass {e_plain:.2e}")
    assert e_fold <= 1.5e-2 and e_fold <= 1.5 * e_plain
    # a weight change invalidates the fold
    sd2 = {k: (v * 1.25 if k == "conv1.weight" else v) for k, v in sd.items()}
    exact.load_state_dict(sd2)
    net.load_state_dict(sd2)
    with torch.no_grad():
        want2 = exact(x)
    got2 = net(x.cuda()).float().cpu()
    assert R.rel_l2(got2, want2) <= 1.5e-2 and R.rel_l2(got2, folded) > R.rel_l2(got2, want2)


def test_three_optimizer_steps_track_oracle():
    """AdamW (per-step lr / wd injection incl. the reference's lr = 0 first step) + EMA + bf16 filter refresh over 3
    steps.  Adam's normalised update turns the chaotic part of a gradient (see the re-association test) into
    O(lr) parameter differences, so the oracle optimizer is driven with the HIP path's OWN gradients: parameters,
    optimizer state and EMA must then agree to fp32 round-off, and the losses (same weights) to bf16 forward noise."""
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.ema import ModelEmaV3
    C = 10
    ref, net = _pair("resnet18", C, seed=7)
    opt_ref = torch.optim.AdamW([{"params": list(ref.parameters()), "weight_decay": 5e-4}], lr=1e-3, weight_decay=0.0)
    opt = create_optimizer("adamw", 1e-3, 5e-4, net)
    ema = ModelEmaV3(net, decay=0.9)
    ema_ref = {k: v.clone() for k, v in ref.state_dict().items()}
    lrs, wds = [0.0, 5e-4, 1e-3], [5e-4, 4.5e-4, 4e-4]
    g = torch.Generator().manual_seed(11)
    for it in range(3):
        x = torch.randn(8, 3, 64, 64, generator=g)
        y = torch.randint(0, C, (8,), generator=g)
        for grp in opt_ref.param_groups:
            grp["lr"], grp["weight_decay"] = lrs[it], wds[it]
        opt.param_groups[0]["lr"], opt.param_groups[0]["weight_decay"] = lrs[it], wds[it]
        ref.train()
        ref_loss = float(torch.nn.functional.cross_entropy(ref(x), y))   # also updates the oracle's BN statistics
        net.train()
        ws = net.pack(x.cuda())
        net.forward_packed(ws)
        hip_loss = _xent_backward(net, ws, y.cuda(), C)
        assert abs(hip_loss - ref_loss) <= 5e-3 * abs(ref_loss), (it, hip_loss, ref_loss)
        for name, p in ref.named_parameters():
            p.grad = net.grad_of(name).reshape(p.shape).clone()
        opt_ref.step()
        opt.step(model_ema=ema)
        for k, v in ref.state_dict().items():
            if v.dtype.is_floating_point:
                ema_ref[k].lerp_(v, 1.0 - 0.9)
            else:
                ema_ref[k].copy_(v)
        torch.cuda.synchronize()
        sd, rsd = net.state_dict(), ref.state_dict()
        for k in ("conv1.weight", "layer2.0.downsample.0.weight", "fc.weight", "fc.bias", "layer3.1.bn2.weight"):
            assert torch.allclose(sd[k], rsd[k], rtol=1e-4, atol=2e-6), (it, k)
        # the bf16 filters the next forward reads are the rounded new parameters (and their transposes)
        p = net.params["layer1.0.conv1.weight"]
        assert torch.equal(net.shadow[p.offset:p.offset + p.numel].float().cpu(),
                           R.bf16_round(net.param_arena[p.offset:p.offset + p.numel].cpu()))
    esd = ema.state_dict()
    for k in ("conv1.weight", "layer2.0.downsample.0.weight", "fc.weight", "layer3.1.bn2.weight"):
        assert torch.allclose(esd[k], ema_ref[k], rtol=1e-4, atol=2e-6), k
    assert R.rel_l2(esd["bn1.running_mean"], ema_ref["bn1.running_mean"]) <= 2e-2
    assert opt.step_count == 3 and net.num_batches_tracked == 3
    conv = net.blocks[0]["convs"][0]
    w = net.shadow[conv.w.offset:conv.w.offset + conv.w.numel].reshape(conv.cout_p, 9, conv.cin_p)
    wt = net.shadow_t[conv.wt_offset:conv.wt_offset + conv.w.numel].reshape(conv.cin_p, 9, conv.cout_p)
    assert torch.equal(wt, w.permute(2, 1, 0))
