"""train_one_epoch / evaluate for the MI355X path: same signatures, return keys and printed lines as the
reference's engine.py (/root/reference/engine.py:10-143 and :145-225), different execution:

  * every per-step operation is a HIP kernel behind the C ABI (include/icamd.h); nothing on the step path
    goes through autograd or ATen;
  * NO per-step host synchronisation: the loss, its finiteness flag, the running accuracy and the per-class
    TP/FP/FN counters live on the device and are read back once per epoch (the reference blocks on
    loss.item(), cuda.synchronize() and 3*num_classes .item() calls per step, engine.py:54,79,84-96);
  * the non-finite-loss rule (engine.py:56-59: print, drop the step, keep going) is enforced on the device:
    the optimizer/EMA/metric kernels are predicated on the flag, and skipped steps are reported at epoch end.

Behavioural notes kept: schedule values injected every micro-step (C.1), the extra
train-mode forward on the un-mixed images for the accuracy under mixup (C.3, on by default, switch off with
ICAMD_CHEAP_MIXUP_ACC=1), loss meter averages pre-division losses (C.6), eval loss averaged per batch and
acc1 per sample (C.7), precision/recall from rank-local counts (C.8), clipping only on the use_amp branch (C.5).
"""
import os
import time

import torch

from . import hip, utils
from .mixup import sample_params

LOG_RING = 4096


class _StepState:
    """Device-resident per-epoch accumulators."""

    def __init__(self, device, num_classes):
        self.num_classes = num_classes
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        self.finite = torch.ones(1, dtype=torch.int32, device=device)
        self.acc = torch.zeros(8, dtype=torch.float64, device=device)
        self.counts = torch.zeros(3, num_classes, dtype=torch.int32, device=device)
        self.log = torch.zeros(2 * LOG_RING, dtype=torch.float32, device=device)
        self.nonfinite_steps = torch.zeros(1, dtype=torch.int32, device=device)

    def reset(self):
        self.acc.zero_()
        self.counts.zero_()
        self.log.fill_(float("nan"))
        self.finite.fill_(1)


def _state(net, device, num_classes):
    key = (str(device), num_classes)
    cache = net.__dict__.setdefault("_step_states", {})
    if key not in cache:
        cache[key] = _StepState(device, num_classes)
    return cache[key]


def _criterion_smoothing(criterion):
    """Label smoothing of whatever criterion object the caller built (reference train.py:256-261): this package's classes
    and timm's LabelSmoothingCrossEntropy carry `.smoothing`, torch.nn.CrossEntropyLoss carries `.label_smoothing`,
    SoftTargetCrossEntropy (mixup) carries neither -- its targets arrive already smoothed by the Mixup object."""
    for attr in ("smoothing", "label_smoothing"):
        v = getattr(criterion, attr, None)
        if isinstance(v, (int, float)):
            return float(v)
    return 0.0


def _unwrap(model):
    return model.module if hasattr(model, "reducer") else model


def _print_class_table(tp, fp, fn, num_classes):
    out = []
    for i in range(num_classes):
        precision = tp[i] / (tp[i] + fp[i]) if tp[i] + fp[i] > 0 else 0
        recall = tp[i] / (tp[i] + fn[i]) if tp[i] + fn[i] > 0 else 0
        print(f"Class {i}: Precision: {precision:.5f}, Recall: {recall:.5f}")
        out.append((precision, recall))
    return out


def train_one_epoch(model, criterion, data_loader, optimizer, device, epoch, loss_scaler, max_norm=0, model_ema=None,
                    mixup_fn=None, log_writer=None, wandb_logger=None, start_steps=None, lr_schedule_values=None,
                    wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=None, use_amp=False,
                    num_classes=2):
    net = _unwrap(model)
    reducer = getattr(model, "reducer", None)
    lib = net.lib
    device = torch.device(device)
    model.train(True)
    metric_logger = utils.MetricLogger(delimiter="  ")
    optimizer.zero_grad()
    start_time = time.time()
    st = _state(net, device, num_classes)
    st.reset()
    update_freq = update_freq or 1
    start_steps = start_steps or 0
    if num_training_steps_per_epoch is None:
        num_training_steps_per_epoch = len(data_loader) // update_freq
    cheap_mixup_acc = os.environ.get("ICAMD_CHEAP_MIXUP_ACC", "0") == "1"
    per_step_host_log = log_writer is not None or bool(wandb_logger)
    grad_scale = reducer.grad_scale if reducer is not None else 1.0
    hook = net.grad_ready_hook
    steps_run = 0
    grad_norm = None
    # data-parallel steps: the finite-loss flag is MIN-reduced over the ranks right after the loss (it depends on the forward
    # only), the metrics are accumulated from the REDUCED flag at the end of the step, and -- when nothing needs the whole
    # gradient first (no clipping, no accumulation window) -- the optimizer runs as one launch per gradient bucket on the
    # reducer's side stream, each behind that bucket's all-reduce (reference: DDP hooks + optimizer.step(), train.py:218-222,
    # engine.py:74).  ICAMD_BUCKET_OPTIM=0 keeps the single launch behind the last bucket.
    dp = reducer is not None and reducer.active and reducer.on_gpu
    bucket_optim = (dp and update_freq == 1 and not (use_amp and max_norm is not None) and hasattr(optimizer, "step_range")
                    and os.environ.get("ICAMD_BUCKET_OPTIM", "1") != "0")
    if reducer is not None:
        reducer.bucket_callback = None

    for data_iter_step, (samples, targets) in enumerate(data_loader):
        step = data_iter_step // update_freq
        if step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        # reference engine.py:33 precedence: injected on every micro-step
        if lr_schedule_values is not None or wd_schedule_values is not None and data_iter_step % update_freq == 0:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]

        samples = samples.to(device, dtype=torch.float32, non_blocking=True).contiguous()
        targets = targets.to(device, dtype=torch.int64, non_blocking=True).contiguous()
        B = samples.shape[0]
        s = hip.stream_ptr()

        # the reference calls mixup_fn(samples, targets) (engine.py:44) on whatever train.py:176-185 built; here the mixing
        # runs inside the packing / loss kernels, so only the object's configuration is used to draw (mode, lambda, box)
        mix = None
        if mixup_fn is not None:
            mix = mixup_fn.sample(samples.shape) if hasattr(mixup_fn, "sample") else sample_params(mixup_fn, samples.shape)
        ws = net.pack(samples, mix)
        logits = net.forward_packed(ws)

        if mixup_fn is not None:
            smoothing, lam = float(mixup_fn.label_smoothing), mix[1]
            flipped = targets.flip(0).contiguous()
        else:
            smoothing, lam, flipped = _criterion_smoothing(criterion), 1.0, None
        slot = steps_run % LOG_RING
        want_pred = mixup_fn is None or cheap_mixup_acc
        hip.check(lib.icamd_softmax_xent(logits.data_ptr(), net.ncls_p, B, num_classes, targets.data_ptr(),
                                         None if flipped is None else flipped.data_ptr(), float(lam), float(smoothing),
                                         1.0 / (B * update_freq), ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(),
                                         ws["dlogits"].data_ptr(), s), "softmax_xent")
        pred_ptr = ws["pred"].data_ptr() if want_pred else None
        is_update = (data_iter_step + 1) % update_freq == 0
        if dp:
            # flag call (loss, flag, log slot; nothing accumulated) -> MIN over the ranks on the side stream
            hip.check(lib.icamd_step_metrics(ws["loss_rows"].data_ptr(), None, targets.data_ptr(), B, num_classes,
                                             st.loss.data_ptr(), st.finite.data_ptr(), st.acc.data_ptr(), st.counts.data_ptr(),
                                             st.log.data_ptr(), slot, LOG_RING, 2, s), "step_metrics (flag)")
            reducer.reduce_flag(st.finite)
            if bucket_optim and is_update:
                optimizer.begin_step()
                reducer.bucket_callback = lambda lo, hi, _bi: optimizer.step_range(
                    lo, hi, model_ema=model_ema, grad_scale=grad_scale, finite_flag=st.finite)
        else:
            hip.check(lib.icamd_step_metrics(ws["loss_rows"].data_ptr(), pred_ptr, targets.data_ptr(), B, num_classes,
                                             st.loss.data_ptr(), st.finite.data_ptr(), st.acc.data_ptr(), st.counts.data_ptr(),
                                             st.log.data_ptr(), slot, LOG_RING, 1, s), "step_metrics")

        net.grad_ready_hook = hook if is_update else None   # reduce once per optimizer step (sum of micro-steps)
        net.backward_packed(ws, accumulate=(data_iter_step % update_freq) != 0)
        net.grad_ready_hook = hook
        if dp:
            if is_update:
                reducer.finish()        # gradients summed over ranks (and, per bucket, applied); main stream waits
                reducer.bucket_callback = None
            else:
                reducer.wait()          # the reduced flag
            # every rank counts (or drops) the same steps: accumulate from the REDUCED flag
            hip.check(lib.icamd_step_metrics(None, pred_ptr, targets.data_ptr(), B, num_classes, st.loss.data_ptr(),
                                             st.finite.data_ptr(), st.acc.data_ptr(), st.counts.data_ptr(), st.log.data_ptr(),
                                             slot, LOG_RING, 1 | 4, s), "step_metrics (accumulate)")
        elif is_update and reducer is not None:
            reducer.finish(st.finite)   # CPU tensors / inactive reducer
        if update_freq > 1:
            # reference engine.py:56-59: a non-finite micro-batch zero_grad()s (dropping what the window had accumulated
            # so far) and is skipped; decided on the device, no host sync
            hip.check(lib.icamd_grad_guard(net.grad_arena.data_ptr(), net.n_params, st.finite.data_ptr(), s), "grad_guard")
        if is_update:
            clip = use_amp and max_norm is not None
            if use_amp:
                # reference utils.py:438-442: clip_grad_norm_ when clip_grad is given, else only measure the norm
                grad_norm = optimizer.measure_grad_norm(max_norm if clip else None, grad_scale=grad_scale)
            if bucket_optim:
                optimizer.finish_step(model_ema, st.finite)     # every range was applied behind its bucket's all-reduce
            else:
                optimizer.step(model_ema=model_ema, grad_scale=grad_scale, use_clip=clip, finite_flag=st.finite)

        if mixup_fn is not None and not cheap_mixup_acc:
            # reference engine.py:89-97: accuracy of the un-mixed images through the (already updated) model,
            # in train mode (BatchNorm statistics are updated a second time)
            ws2 = net.pack(samples, None)
            logits2 = net.forward_packed(ws2, logits_only=True)   # nothing reads its saved activations
            hip.check(lib.icamd_softmax_xent(logits2.data_ptr(), net.ncls_p, B, num_classes, targets.data_ptr(), None, 1.0,
                                             0.0, 0.0, ws2["loss_rows"].data_ptr(), ws2["pred"].data_ptr(), None, s),
                      "argmax")
            hip.check(lib.icamd_step_metrics(None, ws2["pred"].data_ptr(), targets.data_ptr(), B, num_classes,
                                             st.loss.data_ptr(), st.finite.data_ptr(), st.acc.data_ptr(),
                                             st.counts.data_ptr(), st.log.data_ptr(), slot, LOG_RING, 1, s), "step_metrics")
        steps_run += 1

        if per_step_host_log:
            _host_step_log(st, optimizer, log_writer, wandb_logger, use_amp, grad_norm, it, slot)

    # host time the loop needed to ENQUEUE its steps (no device wait happens inside it): bench.py's host_enqueue_ms
    train_one_epoch.last_enqueue_s = time.time() - start_time
    torch.cuda.synchronize()
    end_time = time.time()

    acc = st.acc.cpu().tolist()
    counts = st.counts.cpu().tolist()
    log = st.log.cpu()
    n_ok = int(acc[1])
    if n_ok < steps_run:
        # reference prints this line at the step it happens (engine.py:57); the device-side skip reports at epoch end
        print("Loss is non-finite in {} step(s), stopping training".format(steps_run - n_ok))
    if n_ok > 0:
        loss_meter, acc_meter = metric_logger.meters["loss"], metric_logger.meters["class_acc"]
        loss_meter.total, loss_meter.count = acc[0], n_ok
        acc_meter.total, acc_meter.count = acc[2], n_ok
        first = max(0, steps_run - LOG_RING)
        recent = [i % LOG_RING for i in range(first, steps_run)][-64:]
        for slot in recent:
            lv, av = float(log[slot]), float(log[LOG_RING + slot])
            if lv == lv and abs(lv) != float("inf") and av == av:
                loss_meter.deque.append(lv)
                acc_meter.deque.append(av)
    metric_logger.synchronize_between_processes()
    print(f"Averaged stats:{metric_logger},Time:{end_time - start_time}")
    _print_class_table(counts[0], counts[1], counts[2], num_classes)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def _host_step_log(st, optimizer, log_writer, wandb_logger, use_amp, grad_norm, it, slot):
    """Per-step scalar sinks (reference engine.py:101-132). Costs ONE host sync per step, only when a sink exists."""
    vals = st.log[[slot, LOG_RING + slot]].cpu().tolist()
    loss_value, class_acc = vals
    min_lr, max_lr = 10.0, 0.0
    for group in optimizer.param_groups:
        min_lr = min(min_lr, group["lr"])
        max_lr = max(max_lr, group["lr"])
    weight_decay_value = None
    for group in optimizer.param_groups:
        if group["weight_decay"] > 0:
            weight_decay_value = group["weight_decay"]
    gn = float(grad_norm) if (use_amp and grad_norm is not None) else None
    if log_writer is not None:
        log_writer.update(loss=loss_value, head="loss")
        log_writer.update(class_acc=class_acc, head="loss")
        log_writer.update(lr=max_lr, head="opt")
        log_writer.update(min_lr=min_lr, head="opt")
        log_writer.update(weight_decay=weight_decay_value, head="opt")
        if use_amp:
            log_writer.update(grad_norm=gn, head="opt")
        log_writer.set_step()
    if wandb_logger:
        wandb_logger._wandb.log({"Rank-0 Batch Wise/train_loss": loss_value, "Rank-0 Batch Wise/train_max_lr": max_lr,
                                 "Rank-0 Batch Wise/train_min_lr": min_lr}, commit=False)
        if class_acc:
            wandb_logger._wandb.log({"Rank-0 Batch Wise/train_class_acc": class_acc}, commit=False)
        if use_amp:
            wandb_logger._wandb.log({"Rank-0 Batch Wise/train_grad_norm": gn}, commit=False)
        wandb_logger._wandb.log({"Rank-0 Batch Wise/global_train_step": it})


def evaluate(data_loader, model, device, num_classes, use_amp=False):
    if hasattr(model, "sync_buffers"):
        model.sync_buffers()   # DDP semantics: rank 0's BatchNorm statistics are the ones evaluated
    net = _unwrap(model)
    lib = net.lib
    device = torch.device(device)
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = "Val:"
    precision_meters = [utils.SmoothedValue(window_size=1, fmt="{value:.5f}") for _ in range(num_classes)]
    recall_meters = [utils.SmoothedValue(window_size=1, fmt="{value:.5f}") for _ in range(num_classes)]
    metric_logger.add_meter("avg_precision", utils.SmoothedValue(window_size=1, fmt="{value:.5f}"))
    metric_logger.add_meter("avg_recall", utils.SmoothedValue(window_size=1, fmt="{value:.5f}"))
    model.eval()
    if getattr(net, "shadow_stale", False):
        net.refresh_shadow()
        net.shadow_stale = False
    st = _state(net, device, num_classes)
    st.reset()
    nb = 0
    eval_start = time.time()
    for batch in metric_logger.log_every(data_loader, 0, header):
        images, target = batch[0], batch[-1]
        images = images.to(device, dtype=torch.float32, non_blocking=True).contiguous()
        target = target.to(device, dtype=torch.int64, non_blocking=True).contiguous()
        B = images.shape[0]
        s = hip.stream_ptr()
        ws = net.pack(images, None)
        logits = net.forward_packed(ws)
        hip.check(lib.icamd_softmax_xent(logits.data_ptr(), net.ncls_p, B, num_classes, target.data_ptr(), None, 1.0, 0.0,
                                         0.0, ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(), None, s), "softmax_xent")
        hip.check(lib.icamd_step_metrics(ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(), target.data_ptr(), B,
                                         num_classes, st.loss.data_ptr(), st.finite.data_ptr(), st.acc.data_ptr(),
                                         st.counts.data_ptr(), None, 0, 0, 0, s), "step_metrics")
        nb += 1
    evaluate.last_enqueue_s = time.time() - eval_start
    torch.cuda.synchronize()
    acc = st.acc.cpu().tolist()
    counts = st.counts.cpu().tolist()
    if nb > 0:
        # loss: one update per batch (n=1); acc1 (percent): sample-weighted (reference engine.py:194-196)
        m = metric_logger.meters["loss"]
        m.total, m.count = acc[0], int(acc[1])
        m.deque.append(float(st.loss.item()))
        a = metric_logger.meters["acc1"]
        a.total, a.count = 100.0 * acc[3], int(acc[4])
        a.deque.append(100.0 * acc[3] / max(acc[4], 1.0))
    metric_logger.synchronize_between_processes()

    table = _print_class_table(counts[0], counts[1], counts[2], num_classes)
    for i, (precision, recall) in enumerate(table):
        precision_meters[i].update(precision)
        recall_meters[i].update(recall)
        metric_logger.add_meter(f"precision_{i}", precision_meters[i])
        metric_logger.add_meter(f"recall_{i}", recall_meters[i])
    avg_precision = sum(m.global_avg for m in precision_meters) / len(precision_meters)
    avg_recall = sum(m.global_avg for m in recall_meters) / len(recall_meters)
    metric_logger.meters["avg_precision"].update(avg_precision)
    metric_logger.meters["avg_recall"].update(avg_recall)
    print(f"Average Precision: {avg_precision:.5f}, Average Recall: {avg_recall:.5f}")
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
