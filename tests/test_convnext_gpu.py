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


def test_convnext_forward_backward_matches_oracle():
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
