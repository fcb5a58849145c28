"""GPU tests of the drop-in boundary: imageclassification_amd.engine.train_one_epoch / evaluate against the CPU
oracle of the reference loop (oracle/engine_ref.py, itself pinned to the reference's engine.py by
tests/golden/engine_trace.json).  The learning rate is kept tiny where values are compared so that both sides
evaluate (nearly) the same weights and only bf16 forward noise (<= 5e-3 on a loss) separates them."""
import numpy as np
import pytest
import torch

from oracle import engine_ref as E
from oracle import ops_ref as R
from oracle.resnet_ref import ResNetRef

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _setup(C=10, seed=0, arch="resnet18"):
    from imageclassification_amd.nets import ResNet
    from imageclassification_amd.optim_factory import create_optimizer
    torch.manual_seed(seed)
    ref = ResNetRef(arch, C, bf16_points=True, zero_init_last=False)
    net = ResNet(arch, C)
    net.load_state_dict(ref.state_dict())
    opt = create_optimizer("adamw", 1e-6, 5e-4, net)
    opt_ref = torch.optim.AdamW([{"params": list(ref.parameters()), "weight_decay": 5e-4}], lr=1e-6, weight_decay=0.0)
    return ref, net, opt, opt_ref


def _loader(n, B, C, seed, hw=64):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 3, hw, hw, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(n)]


def _train(net, opt, loader, C, crit=None, mixup_fn=None, ema=None, update_freq=1, steps=None, lr=None, wd=None,
           use_amp=True, max_norm=None):
    from imageclassification_amd.engine import train_one_epoch
    from imageclassification_amd.mixup import LabelSmoothingCrossEntropy
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    steps = steps if steps is not None else len(loader) // update_freq
    return train_one_epoch(net, crit or LabelSmoothingCrossEntropy(0.1), loader, opt, DEV, 0,
                           NativeScalerWithGradNormCount(), max_norm, ema, mixup_fn, start_steps=0,
                           lr_schedule_values=lr if lr is not None else [1e-6] * steps,
                           wd_schedule_values=wd if wd is not None else [5e-4] * steps,
                           num_training_steps_per_epoch=steps, update_freq=update_freq, use_amp=use_amp, num_classes=C)


def test_train_one_epoch_matches_oracle_loop():
    C, B = 10, 8
    ref, net, opt, opt_ref = _setup(C)
    data = _loader(4, B, C, seed=21)
    trace = []
    rstats = E.train_one_epoch_ref(ref, E.LabelSmoothingCrossEntropyRef(0.1), [(x.clone(), y.clone()) for x, y in data],
                                   opt_ref, lr_schedule_values=[1e-6] * 4, wd_schedule_values=[5e-4] * 4,
                                   num_training_steps_per_epoch=4, num_classes=C, trace=trace)
    stats = _train(net, opt, data, C)
    assert list(stats) == ["loss", "class_acc"]
    assert abs(stats["loss"] - rstats["loss"]) <= 5e-3 * rstats["loss"]
    assert abs(stats["class_acc"] - rstats["class_acc"]) <= 2.0 / B      # argmax of near-tied bf16 logits may differ
    assert opt.step_count == 4 and net.num_batches_tracked == 4
    # schedule injection reached the optimizer (reference engine.py:33-38)
    assert opt.param_groups[0]["lr"] == 1e-6 and opt.param_groups[0]["weight_decay"] == 5e-4
    # per-class counters: every sample lands in exactly one of TP[t] / FN[t]
    st = list(net._step_states.values())[0]
    counts = st.counts.cpu()
    assert int(counts[0].sum() + counts[2].sum()) == 4 * B and int(counts[1].sum()) == int(counts[2].sum())
    assert int(counts[0].sum()) == round(stats["class_acc"] * B * 4)


def test_config0_at_its_own_size_one_step_matches_oracle():
    """BASELINE configs[0] at ITS OWN size: ResNet-18, 2 classes, batch 32 at 224 x 224 (the reference's
    cat/dog ImageFolder case, /root/reference/train.py:36 --input_size 224, README.md:19-21), one optimizer step of the reference
    recipe through train_one_epoch and an evaluate() pass over two batches -- 32 and a ragged 19 -- against the oracle loop."""
    from imageclassification_amd.engine import evaluate
    C, B, HW = 2, 32, 224
    ref, net, opt, opt_ref = _setup(C, seed=9)
    data = _loader(1, B, C, seed=31, hw=HW)
    rstats = E.train_one_epoch_ref(ref, E.LabelSmoothingCrossEntropyRef(0.1), [(x.clone(), y.clone()) for x, y in data], opt_ref,
                                   lr_schedule_values=[1e-6], wd_schedule_values=[5e-4], num_training_steps_per_epoch=1,
                                   num_classes=C)
    stats = _train(net, opt, data, C)
    assert list(stats) == ["loss", "class_acc"]
    assert abs(stats["loss"] - rstats["loss"]) <= 5e-3 * rstats["loss"], (stats, rstats)
    assert abs(stats["class_acc"] - rstats["class_acc"]) <= 2.0 / B
    assert opt.steps_taken == 1 and net.num_batches_tracked == 1
    g = torch.Generator().manual_seed(32)
    val = [(torch.randn(n, 3, HW, HW, generator=g), torch.randint(0, C, (n,), generator=g)) for n in (B, 19)]
    rev = E.evaluate_ref([(x.clone(), y.clone()) for x, y in val], ref, C)
    ev = evaluate(val, net, DEV, C)
    assert set(ev) == set(rev) and {"loss", "acc1", "avg_precision", "avg_recall", "precision_1", "recall_1"} <= set(ev)
    assert abs(ev["loss"] - rev["loss"]) <= 5e-3 * rev["loss"] and abs(ev["acc1"] - rev["acc1"]) <= 100.0 * 2 / (B + 19)


def test_non_finite_loss_skips_the_step_on_device():
    C, B = 10, 8
    _, net, opt, _ = _setup(C, seed=1)
    data = _loader(3, B, C, seed=22)
    before = net.param_arena.clone()
    bad = [(data[0][0].clone(), data[0][1])]
    bad[0][0][0, 0, 0, 0] = float("inf")
    stats = _train(net, opt, bad, C, lr=[1e-2], wd=[0.0])
    assert stats == {}                                       # no meter was updated (reference: `continue`)
    assert torch.equal(net.param_arena, before)              # AdamW and EMA were predicated off
    assert opt.exp_avg.abs().max().item() == 0.0


def test_non_finite_micro_batch_under_update_freq_zeroes_the_window():
    """update_freq=2 with an inf in the FIRST micro-batch: the reference zero_grad()s and `continue`s
    (engine.py:56-59), so the step that closes the window applies the second micro-batch's gradient only -- and nothing
    may become NaN.  Then an inf in the LAST micro-batch: no step at all, gradients cleared, Adam's step state untouched."""
    C, B = 10, 8
    _, net, opt, _ = _setup(C, seed=4)
    from imageclassification_amd import hip
    data = _loader(2, B, C, seed=25)
    bad0 = (data[0][0].clone(), data[0][1])
    bad0[0][0, 0, 0, 0] = float("inf")
    # expected gradient: micro-batch 1 alone, scaled 1/(B*2)
    net.train()
    x, y = data[1]
    ws = net.pack(x.cuda())
    net.forward_packed(ws)
    hip.check(net.lib.icamd_softmax_xent(ws["logits"].data_ptr(), net.ncls_p, B, C, y.cuda().data_ptr(), None, 1.0, 0.1,
                                         1.0 / (B * 2), ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(),
                                         ws["dlogits"].data_ptr(), hip.stream_ptr()), "xent")
    net.backward_packed(ws, accumulate=False)
    torch.cuda.synchronize()
    expect = net.grad_arena.clone()
    before = net.param_arena.clone()
    stats = _train(net, opt, [bad0, data[1]], C, update_freq=2, lr=[0.0], wd=[0.0])
    got = net.grad_arena.clone()
    assert torch.isfinite(got).all() and torch.isfinite(net.param_arena).all() and torch.isfinite(opt.exp_avg).all()
    assert R.rel_l2(got.cpu(), expect.cpu()) <= 2e-2          # BN batch statistics are identical; bf16 noise only
    assert opt.steps_taken == 1 and abs(stats["loss"]) > 0    # the window's step happened, on the finite micro-batch
    assert torch.equal(net.param_arena, before)               # lr = 0
    # inf in the micro-batch that closes the window: the whole step is dropped and the arena cleared
    m_before = opt.exp_avg.clone()
    bad1 = (data[1][0].clone(), data[1][1])
    bad1[0][0, 0, 0, 0] = float("inf")
    _train(net, opt, [data[0], bad1], C, update_freq=2, lr=[1e-2], wd=[0.0])
    assert torch.equal(net.param_arena, before) and torch.equal(opt.exp_avg, m_before)
    assert float(net.grad_arena.abs().max()) == 0.0
    assert opt.step_count == 2 and opt.steps_taken == 1       # attempted twice, applied once (bias correction uses 1)


def test_reference_criterion_objects_are_accepted():
    """A caller that builds its criterion the reference's way (train.py:256-261: torch.nn.CrossEntropyLoss, or timm's
    LabelSmoothingCrossEntropy, which carries `.smoothing`) gets the same step as with this package's criterion classes."""
    C, B = 10, 8
    data = _loader(1, B, C, seed=31)
    losses = []
    for crit in (None, torch.nn.CrossEntropyLoss(label_smoothing=0.1), torch.nn.CrossEntropyLoss()):
        _, net, opt, _ = _setup(C, seed=6)
        losses.append(_train(net, opt, data, C, crit=crit)["loss"])
    assert losses[0] == losses[1] and losses[2] != losses[0]


def test_update_freq_accumulates_micro_batches():
    """update_freq=2: the gradient applied is d(loss_a/2 + loss_b/2) (reference engine.py:71-72)."""
    C, B = 10, 8
    _, net, opt, _ = _setup(C, seed=2)
    from imageclassification_amd import hip
    data = _loader(2, B, C, seed=23)
    net.train()
    grads = []
    for x, y in data:
        ws = net.pack(x.cuda())
        net.forward_packed(ws)
        hip.check(net.lib.icamd_softmax_xent(ws["logits"].data_ptr(), net.ncls_p, B, C, y.cuda().data_ptr(), None, 1.0, 0.1,
                                             1.0 / (B * 2), ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(),
                                             ws["dlogits"].data_ptr(), hip.stream_ptr()), "xent")
        net.backward_packed(ws, accumulate=False)
        torch.cuda.synchronize()
        grads.append(net.grad_arena.clone())
    expect = grads[0] + grads[1]
    # same two micro-batches through the engine with an lr of 0 so the weights (hence gradients) are unchanged
    sd = net.state_dict()
    for k in list(sd):
        if "running" in k:
            sd[k] = torch.zeros_like(sd[k]) if "mean" in k else torch.ones_like(sd[k])
    stats = _train(net, opt, data, C, update_freq=2, lr=[0.0], wd=[0.0])
    assert opt.step_count == 1 and abs(stats["loss"]) > 0
    got = net.grad_arena.clone()
    assert R.rel_l2(got.cpu(), expect.cpu()) <= 2e-2      # second run sees BN batch stats only: same up to bf16 noise


def test_mixup_path_runs_second_forward_and_soft_targets():
    from imageclassification_amd.mixup import Mixup, SoftTargetCrossEntropy
    C, B = 10, 8
    _, net, opt, _ = _setup(C, seed=3)
    np.random.seed(3)
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
    data = _loader(3, B, C, seed=24)
    stats = _train(net, opt, data, C, crit=SoftTargetCrossEntropy(), mixup_fn=mix)
    assert set(stats) == {"loss", "class_acc"} and np.isfinite(stats["loss"])
    assert net.num_batches_tracked == 6      # reference engine.py:90-91: a second train-mode forward per step
    # soft-target loss of a mixed batch is at least the entropy floor of the target distribution
    assert stats["loss"] > 0.5


class _TimmStyleMixup:
    """Stand-in for the object the reference builds (timm.data.Mixup, train.py:176-185): timm's attribute names and a
    __call__(x, target) -- which the MI355X engine never calls (the mixing runs in the packing / loss kernels)."""

    def __init__(self, mixup_alpha=1., cutmix_alpha=0., cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        self.mix_prob, self.switch_prob, self.label_smoothing, self.num_classes = prob, switch_prob, label_smoothing, num_classes
        self.mode, self.correct_lam, self.mixup_enabled = mode, correct_lam, True

    def __call__(self, x, target):
        raise AssertionError("the engine must not run the host-side mixing")


def test_reference_mixup_object_is_accepted():
    """`mixup_fn` as the reference passes it (engine.py:44): an object with timm.data.Mixup's public fields gives the very
    step this package's Mixup gives (same numpy draws, same kernels); unsupported modes raise a clear error."""
    from imageclassification_amd.mixup import Mixup, SoftTargetCrossEntropy
    C, B = 10, 8
    data = _loader(3, B, C, seed=24)
    out = []
    for cls in (Mixup, _TimmStyleMixup):
        _, net, opt, _ = _setup(C, seed=3)
        np.random.seed(3)
        mix = cls(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)
        stats = _train(net, opt, data, C, crit=SoftTargetCrossEntropy(), mixup_fn=mix)
        out.append((stats, net.param_arena.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
    _, net, opt, _ = _setup(C, seed=3)
    with pytest.raises(ValueError, match="mode"):
        _train(net, opt, data, C, crit=SoftTargetCrossEntropy(), mixup_fn=_TimmStyleMixup(mode="elem", num_classes=C))
    with pytest.raises(TypeError, match="mixup_alpha"):
        _train(net, opt, data, C, crit=SoftTargetCrossEntropy(), mixup_fn=object())


def test_evaluate_matches_oracle_and_key_order():
    from imageclassification_amd.engine import evaluate
    C, B = 5, 12
    ref, net, _, _ = _setup(C, seed=4)
    data = _loader(3, B, C, seed=25)
    data[-1] = (data[-1][0][:5], data[-1][1][:5])      # ragged last batch (loss is per batch, acc1 per sample)
    rev = E.evaluate_ref(data, ref, C)
    ev = evaluate(data, net, DEV, C)
    assert list(ev) == list(rev)
    assert abs(ev["loss"] - rev["loss"]) <= 5e-3 * abs(rev["loss"])
    assert abs(ev["acc1"] - rev["acc1"]) <= 100.0 * 2 / 29 + 1e-6
    st = list(net._step_states.values())[0]
    counts = st.counts.cpu()
    assert int(counts[0].sum() + counts[2].sum()) == 29
    assert net.num_batches_tracked == 0 and not net.training


def test_grad_clip_and_ema_in_engine():
    from imageclassification_amd.ema import ModelEmaV3
    C, B = 10, 8
    _, net, opt, _ = _setup(C, seed=5)
    ema = ModelEmaV3(net, decay=0.5)
    p0 = net.param_arena.clone()
    data = _loader(1, B, C, seed=26)
    _train(net, opt, data, C, ema=ema, lr=[1e-3], wd=[0.0], max_norm=1e-3)
    torch.cuda.synchronize()
    norm, coef = opt.norm_clip.cpu().tolist()
    assert norm > 1e-3 and abs(coef - 1e-3 / (norm + 1e-6)) <= 1e-6 * coef + 1e-9
    moved = (net.param_arena - p0).abs().max().item()
    assert 0 < moved <= 1.01e-3                          # first Adam step moves every weight by at most lr
    assert torch.allclose(ema.param_arena, p0 + 0.5 * (net.param_arena - p0), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["sgd", "lion"])
def test_train_one_epoch_with_the_other_fused_optimizers(name):
    """The reference's `--opt sgd|lion` recipes through the same boundary (optim_factory.py:66-77): losses track the
    oracle loop and the classifier's update has the oracle's direction and size."""
    from imageclassification_amd.optim_factory import create_optimizer
    C, B, steps = 10, 8, 3
    ref, net, _, _ = _setup(C)
    opt = create_optimizer(name, 1e-4, 5e-4, net)
    groups = [{"params": list(ref.parameters()), "weight_decay": 5e-4}]
    opt_ref = (torch.optim.SGD(groups, lr=1e-4, momentum=0.9, nesterov=True, weight_decay=0.0) if name == "sgd"
               else R.LionRef(groups, betas=(0.9, 0.999)))
    data = _loader(steps, B, C, seed=23)
    fc0 = ref.state_dict()["fc.weight"].clone()
    lrs = [1e-4, 2e-4, 3e-4] if name == "sgd" else [1e-6] * steps
    rstats = E.train_one_epoch_ref(ref, E.LabelSmoothingCrossEntropyRef(0.1), [(x.clone(), y.clone()) for x, y in data],
                                   opt_ref, lr_schedule_values=lrs, wd_schedule_values=[5e-4] * steps,
                                   num_training_steps_per_epoch=steps, num_classes=C)
    stats = _train(net, opt, data, C, lr=lrs, wd=[5e-4] * steps)
    assert abs(stats["loss"] - rstats["loss"]) <= 5e-3 * rstats["loss"]
    assert opt.step_count == steps and opt.exp_avg_sq is None
    d_ref = ref.state_dict()["fc.weight"] - fc0
    d_hip = net.state_dict()["fc.weight"].cpu().float() - fc0
    if name == "sgd":
        assert R.rel_l2(d_hip, d_ref) <= 5e-2
    else:   # sign updates: every weight moves by lr per step; the sign pattern agrees except where |update| ~ 0
        agree = (torch.sign(d_hip) == torch.sign(d_ref)).float().mean()
        assert agree >= 0.97
