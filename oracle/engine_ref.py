"""CPU oracle of the step engine (TEST INFRASTRUCTURE ONLY -- never imported by the product).

A torch-CPU restatement of /root/reference/engine.py:train_one_epoch (:10-143) and evaluate (:145-225) with the
collaborators the reference takes from timm restated next to it (timm is absent here: Mixup, the two soft/
smoothed criteria, ModelEmaV3 and `accuracy` are written from their published behaviour -- [recall], pinned
only against torch primitives and, for the loop itself, against the reference's own engine.py executed in
this container with stand-ins for the missing names (tests/golden/make_engine_fixture.py)).

Differences from the reference file, all forced by running without a GPU: `torch.cuda.synchronize()`
(engine.py:79) is dropped and the progress bar (engine.py:24-28) is not drawn.  Everything else keeps the
reference's order of operations, including the quirks listed in SURVEY.md Appendix C.
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- timm restatements [recall]
class MixupRef:
    """timm.data.Mixup, 'batch' mode. In-place on x like timm (x.mul_(lam).add_(x.flip(0)*(1-lam)))."""

    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        if cutmix_minmax is not None:
            self.cutmix_alpha = 1.0
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes, self.correct_lam = label_smoothing, num_classes, correct_lam
        self.last = None  # (mode, lam, box) of the most recent call, for parity tests that inject the draw

    def _params(self):
        lam, use_cutmix = 1.0, False
        if np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand() < self.switch_prob
                a = self.cutmix_alpha if use_cutmix else self.mixup_alpha
                lam = float(np.random.beta(a, a))
            elif self.mixup_alpha > 0.0:
                lam = float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam = float(np.random.beta(self.cutmix_alpha, self.cutmix_alpha))
        return lam, use_cutmix

    def __call__(self, x, target):
        assert len(x) % 2 == 0, "Batch size should be even when using this"
        lam, use_cutmix = self._params()
        mode, box = 0, (0, 0, 0, 0)
        if lam != 1.0:
            if use_cutmix:
                H, W = x.shape[-2:]
                ratio = np.sqrt(1 - lam)
                cut_h, cut_w = int(H * ratio), int(W * ratio)
                cy, cx = np.random.randint(0, H), np.random.randint(0, W)
                yl, yh = int(np.clip(cy - cut_h // 2, 0, H)), int(np.clip(cy + cut_h // 2, 0, H))
                xl, xh = int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cx + cut_w // 2, 0, W))
                if self.correct_lam:
                    lam = 1.0 - (yh - yl) * (xh - xl) / float(H * W)
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
                mode, box = 2, (yl, yh, xl, xh)
            else:
                x_flipped = x.flip(0).mul_(1.0 - lam)
                x.mul_(lam).add_(x_flipped)
                mode = 1
        self.last = (mode, lam, box)
        off = self.label_smoothing / self.num_classes
        on = 1.0 - self.label_smoothing + off
        y1 = torch.full((target.shape[0], self.num_classes), off).scatter_(1, target.view(-1, 1), on)
        y2 = torch.full((target.shape[0], self.num_classes), off).scatter_(1, target.flip(0).view(-1, 1), on)
        return x, y1 * lam + y2 * (1.0 - lam)


class SoftTargetCrossEntropyRef(torch.nn.Module):
    def forward(self, x, target):
        return torch.sum(-target * F.log_softmax(x, dim=-1), dim=-1).mean()


class LabelSmoothingCrossEntropyRef(torch.nn.Module):
    def __init__(self, smoothing=0.1):
        super().__init__()
        self.smoothing, self.confidence = smoothing, 1.0 - smoothing

    def forward(self, x, target):
        logprobs = F.log_softmax(x, dim=-1)
        nll = -logprobs.gather(dim=-1, index=target.unsqueeze(1)).squeeze(1)
        smooth = -logprobs.mean(dim=-1)
        return (self.confidence * nll + self.smoothing * smooth).mean()


class ModelEmaRef:
    """timm.utils.ModelEmaV3 with constant decay: lerp of every float state_dict entry, copy of the others."""

    def __init__(self, model, decay=0.9999):
        import copy
        self.module = copy.deepcopy(model).eval()
        self.decay = decay

    @torch.no_grad()
    def update(self, model):
        for e, m in zip(self.module.state_dict().values(), model.state_dict().values()):
            if e.is_floating_point():
                e.lerp_(m.to(e.dtype), 1.0 - self.decay)
            else:
                e.copy_(m)


def accuracy_ref(output, target, topk=(1,)):
    maxk = min(max(topk), output.size(1))
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.size(0) for k in topk]


# ---------------------------------------------------------------- the loop (reference engine.py)
class _Meter:
    def __init__(self):
        self.total, self.count, self.last = 0.0, 0, None

    def update(self, v, n=1):
        v = float(v)
        self.total += v * n
        self.count += n
        self.last = v

    @property
    def global_avg(self):
        return self.total / self.count


def train_one_epoch_ref(model, criterion, data_loader, optimizer, epoch=0, max_norm=None, model_ema=None, mixup_fn=None,
                        start_steps=0, lr_schedule_values=None, wd_schedule_values=None,
                        num_training_steps_per_epoch=None, update_freq=1, num_classes=2, trace=None, cpu_alias=False):
    """fp32 branch of the reference (use_amp=False, engine.py:70-77: no clipping). Returns the reference's dict
    plus, in `trace` (a list), one record per executed step: loss, class_acc, cumulative TP/FP/FN.
    cpu_alias=True reproduces what the reference does when device == cpu: `.to(device)` returns the SAME tensor,
    so `original_samples` aliases the in-place-mixed batch (SURVEY Appx C.2); on a GPU they are separate copies."""
    model.train(True)
    meters = {}
    optimizer.zero_grad()
    tp, fp, fn = [0] * num_classes, [0] * num_classes, [0] * num_classes
    for data_iter_step, (samples, targets) in enumerate(data_loader):
        step = data_iter_step // update_freq
        if step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None and data_iter_step % update_freq == 0:
            for group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    group["lr"] = lr_schedule_values[it]
                if wd_schedule_values is not None and group["weight_decay"] > 0:
                    group["weight_decay"] = wd_schedule_values[it]
        # engine.py:40-41: on a GPU these are two separate device copies, so the "original" stays un-mixed
        if cpu_alias:
            original_samples, original_targets = samples, targets
        else:
            original_samples, original_targets = samples.clone(), targets.clone()
            samples = samples.clone()
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        output = model(samples)
        loss = criterion(output, targets)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            optimizer.zero_grad()
            continue
        loss = loss / update_freq
        loss.backward()
        if (data_iter_step + 1) % update_freq == 0:
            optimizer.step()
            optimizer.zero_grad()
            if model_ema is not None:
                model_ema.update(model)
        if mixup_fn is None:
            preds = output.argmax(1)
            ref_t = targets
        else:
            with torch.no_grad():
                preds = model(original_samples).argmax(1)
            ref_t = original_targets
        for i in range(num_classes):
            tp[i] += int(((preds == i) & (ref_t == i)).sum())
            fp[i] += int(((preds == i) & (ref_t != i)).sum())
            fn[i] += int(((preds != i) & (ref_t == i)).sum())
        class_acc = (preds == ref_t).float().mean().item()
        meters.setdefault("loss", _Meter()).update(loss_value)
        meters.setdefault("class_acc", _Meter()).update(class_acc)
        if trace is not None:
            trace.append({"loss": loss_value, "class_acc": class_acc, "tp": list(tp), "fp": list(fp), "fn": list(fn)})
    return {k: m.global_avg for k, m in meters.items()}


@torch.no_grad()
def evaluate_ref(data_loader, model, num_classes):
    tp, fp, fn = [0] * num_classes, [0] * num_classes, [0] * num_classes
    loss_m, acc_m = _Meter(), _Meter()
    model.eval()
    for batch in data_loader:
        images, target = batch[0], batch[-1]
        output = model(images)
        loss = F.cross_entropy(output, target)
        preds = output.argmax(1)
        for i in range(num_classes):
            tp[i] += int(((preds == i) & (target == i)).sum())
            fp[i] += int(((preds == i) & (target != i)).sum())
            fn[i] += int(((preds != i) & (target == i)).sum())
        acc1 = accuracy_ref(output, target, topk=(1,))[0]
        loss_m.update(loss.item())
        acc_m.update(acc1.item(), n=images.shape[0])
    out = {}
    precs, recs = [], []
    for i in range(num_classes):
        precs.append(tp[i] / (tp[i] + fp[i]) if tp[i] + fp[i] > 0 else 0)
        recs.append(tp[i] / (tp[i] + fn[i]) if tp[i] + fn[i] > 0 else 0)
    out["avg_precision"] = sum(precs) / len(precs)
    out["avg_recall"] = sum(recs) / len(recs)
    out["loss"] = loss_m.global_avg
    out["acc1"] = acc_m.global_avg
    for i in range(num_classes):
        out[f"precision_{i}"] = precs[i]
        out[f"recall_{i}"] = recs[i]
    return out


def cosine_scheduler_ref(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0,
                         warmup_steps=-1):
    """reference utils.py:471-488"""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / (len(iters))))
                         for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def time_cpu_training(arch="resnet50", batch=32, hw=224, num_classes=1000, warmup=1, steps=3, threads=None, seed=88):
    """CPU baseline for bench.py: the restated step (fp32, AdamW, label smoothing 0.1, cosine lr/wd injected per
    step) on synthetic data.  Returns (images_per_second, threads_used, seconds_per_step)."""
    from .resnet_ref import ResNetRef
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    model = ResNetRef(arch, num_classes)
    opt = torch.optim.AdamW([{"params": list(model.parameters()), "weight_decay": 5e-4}], lr=1e-3, weight_decay=0.0)
    crit = LabelSmoothingCrossEntropyRef(0.1)
    n = warmup + steps
    lr = cosine_scheduler_ref(1e-3, 1e-6, 1, n, warmup_epochs=0)
    wd = cosine_scheduler_ref(5e-4, 5e-6, 1, n)
    data = [(torch.randn(batch, 3, hw, hw, generator=g), torch.randint(0, num_classes, (batch,), generator=g))
            for _ in range(2)]
    loader = [data[i % 2] for i in range(n)]
    train_one_epoch_ref(model, crit, loader[:warmup], opt, lr_schedule_values=lr, wd_schedule_values=wd,
                        num_training_steps_per_epoch=warmup, num_classes=num_classes) if warmup else None
    t0 = time.time()
    train_one_epoch_ref(model, crit, loader[warmup:], opt, start_steps=warmup, lr_schedule_values=lr,
                        wd_schedule_values=wd, num_training_steps_per_epoch=steps, num_classes=num_classes)
    dt = time.time() - t0
    return batch * steps / dt, torch.get_num_threads(), dt / steps


def time_cpu_eval(arch="resnet50", batch=32, hw=224, num_classes=1000, warmup=1, steps=3, threads=None, seed=88):
    """CPU baseline for `bench.py --mode eval`: the restated evaluate loop (reference engine.py:145-225: eval-mode forward,
    cross-entropy, top-1, per-class counts) on synthetic data, fp32.  Returns (images_per_second, threads_used, seconds_per_batch)."""
    from .resnet_ref import ResNetRef
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    model = ResNetRef(arch, num_classes)
    data = [(torch.randn(batch, 3, hw, hw, generator=g), torch.randint(0, num_classes, (batch,), generator=g)) for _ in range(2)]
    with torch.no_grad():
        if warmup:
            evaluate_ref([data[i % 2] for i in range(warmup)], model, num_classes)
        t0 = time.time()
        evaluate_ref([data[i % 2] for i in range(steps)], model, num_classes)
        dt = time.time() - t0
    return batch * steps / dt, torch.get_num_threads(), dt / steps
