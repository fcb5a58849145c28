"""Optimizer construction for the HIP step (reference surface: /root/reference/optim_factory.py:23-122).

The reference puts EVERY trainable parameter (biases and BatchNorm affine terms included) in one "decay"
group with weight_decay = wd and forces the optimizer-level weight decay to 0 (optim_factory.py:23-55), then
builds torch.optim.AdamW with default betas/eps (`--opt_eps/--opt_betas` are parsed but never forwarded,
train.py:224-229).  Here that is one fused kernel over the model's flat fp32 arenas (icamd_adamw_ema), which
also applies the ModelEmaV3 lerp and writes the bf16 filters the conv kernels read.
"""
import torch

from . import hip


def get_parameter_groups(model, weight_decay=1e-5):
    """One group named "decay" holding every parameter (reference optim_factory.py:23-47)."""
    return [{"weight_decay": weight_decay, "params": list(model.parameters()), "name": "decay"}]


class FusedAdamW:
    """torch.optim.AdamW semantics (decoupled decay, bias correction, amsgrad off) on flat arenas.

    Protocol used by train_one_epoch (reference engine.py:33-38,74-75,101-110): `param_groups[i]["lr"]`,
    `["weight_decay"]` are rewritten every micro-step from the schedule arrays; `step()`; `zero_grad()`.
    """

    def __init__(self, model, lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.lib = hip.load()
        self.param_groups = [{"lr": lr, "weight_decay": weight_decay, "betas": betas, "eps": eps, "name": "decay",
                              "params": [0]}]
        n = model.n_params
        dev = model.param_arena.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0       # optimizer steps ATTEMPTED (host); the device counts the ones it dropped
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self._gn_ws = torch.empty(self.lib.icamd_grad_norm_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.norm_clip = torch.ones(2, dtype=torch.float32, device=dev)  # [grad norm, clip coefficient]

    # -- the pieces of NativeScaler.__call__ that touch gradients (reference utils.py:438-442,456-468)
    def measure_grad_norm(self, max_norm=None, grad_scale=1.0):
        """Global L2 norm of all gradients (x grad_scale) -> norm_clip[0]; norm_clip[1] = clip coefficient
        min(1, max_norm/(norm+1e-6)) when max_norm is given, else 1."""
        m = self.model
        hip.check(self.lib.icamd_grad_norm(m.grad_arena.data_ptr(), m.n_params, float(grad_scale),
                                           float(max_norm) if max_norm else 0.0, self._gn_ws.data_ptr(),
                                           self.norm_clip.data_ptr(), hip.stream_ptr()), "grad_norm")
        return self.norm_clip[0]

    def _launch(self, lo, hi, model_ema, grad_scale, use_clip, finite_flag, flags):
        """The fused kernel over arena elements [lo, hi) on the current stream."""
        g = self.param_groups[0]
        m = self.model
        ema_ptr = None
        decay = 0.0
        if model_ema is not None:
            ema_ptr = model_ema.param_arena.data_ptr() + 4 * lo
            decay = model_ema.decay
        hip.check(self.lib.icamd_adamw_ema(m.param_arena.data_ptr() + 4 * lo, m.grad_arena.data_ptr() + 4 * lo,
                                           self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, ema_ptr,
                                           m.shadow.data_ptr() + 2 * lo, hi - lo,
                                           float(g["lr"]), float(g["weight_decay"]), float(g["betas"][0]),
                                           float(g["betas"][1]), float(g["eps"]), self.step_count, float(grad_scale),
                                           float(decay), self.norm_clip.data_ptr() if use_clip else None,
                                           None if finite_flag is None else finite_flag.data_ptr(),
                                           self.skipped.data_ptr(), int(flags), hip.stream_ptr()), "adamw_ema")

    def step(self, model_ema=None, grad_scale=1.0, use_clip=False, finite_flag=None, zero_grad=False):
        self.step_count += 1
        self._launch(0, self.model.n_params, model_ema, grad_scale, use_clip, finite_flag, int(bool(zero_grad)))
        self.finish_step(model_ema, finite_flag)

    # -- one optimizer step as several launches, one per gradient bucket (ddp.GradReducer.bucket_callback) ------------
    def begin_step(self):
        """Count the step once; the range launches that follow belong to it."""
        self.step_count += 1
        self._ranges_in_step = 0

    def step_range(self, lo, hi, model_ema=None, grad_scale=1.0, finite_flag=None):
        """Apply the step to arena elements [lo, hi) (multiples of 4) on the CURRENT stream -- the reducer's side stream,
        right behind that bucket's all-reduce.  Element for element the same arithmetic as step(): the result is
        bit-identical to the single launch.  Only the first range of a step counts a dropped step."""
        flags = 0 if self._ranges_in_step == 0 else 2      # ICAMD_OPT_NO_SKIP_COUNT
        self._ranges_in_step += 1
        self._launch(lo, hi, model_ema, grad_scale, False, finite_flag, flags)

    def finish_step(self, model_ema=None, finite_flag=None):
        """What follows the parameter update once every range is applied: transposed filters, EMA of the buffers."""
        self.model.refresh_transposed()
        if model_ema is not None:
            model_ema.after_fused_update(self.model, finite_flag)

    @property
    def steps_taken(self):
        """torch.optim's `step` state: steps really applied (steps dropped for a non-finite loss do not count).  Reads the
        device counter (one sync): checkpoint / test use only."""
        return self.step_count - int(self.skipped.item())

    def _load_step(self, step):
        self.step_count = int(step)
        self.skipped.zero_()

    def zero_grad(self, set_to_none=True):
        # gradients are overwritten (not accumulated) by the next backward unless update_freq > 1
        self.model.grad_arena.zero_()

    def state_dict(self):
        return {"state": {"step": self.steps_taken, "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        st = sd["state"]
        self._load_step(st["step"])
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update(src)


OPT_ADAMW, OPT_ADAM, OPT_SGD_MOMENTUM, OPT_SGD_NESTEROV, OPT_LION = range(5)   # include/icamd.h ICAMD_OPT_*


class FusedOptimizer(FusedAdamW):
    """The reference's other optimizers that have a one-pass fused form (optim_factory.py:66-77), on the same flat
    arenas and with the same EMA / shadow / skip fusion as FusedAdamW (icamd_optim_ema):

      sgd | nesterov -> torch.optim.SGD(momentum=0.9, nesterov=True)      momentum -> nesterov=False
      adam           -> torch.optim.Adam (weight decay joins the gradient)
      lion           -> timm Lion(betas=(0.9, 0.999)); the reference does not forward `lr` to it (optim_factory.py:77),
                        which is invisible in training because engine.py:33-38 rewrites lr every step.
    """

    def __init__(self, model, kind, lr=1e-3, weight_decay=0.0, momentum=0.9, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(model, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        self.kind = kind
        g = self.param_groups[0]
        if kind in (OPT_SGD_MOMENTUM, OPT_SGD_NESTEROV):
            g.pop("betas"); g.pop("eps")
            g.update(momentum=momentum, nesterov=kind == OPT_SGD_NESTEROV, dampening=0)
        if kind == OPT_LION:
            g.pop("eps")
        if kind not in (OPT_ADAMW, OPT_ADAM):
            self.exp_avg_sq = None   # only the Adam family keeps second moments

    def _launch(self, lo, hi, model_ema, grad_scale, use_clip, finite_flag, flags):
        g = self.param_groups[0]
        m = self.model
        if self.kind in (OPT_SGD_MOMENTUM, OPT_SGD_NESTEROV):
            b1, b2, eps = g["momentum"], 0.0, 0.0
        else:
            b1, b2, eps = g["betas"][0], g["betas"][1], g.get("eps", 0.0)
        ema_ptr, decay = None, 0.0
        if model_ema is not None:
            ema_ptr, decay = model_ema.param_arena.data_ptr() + 4 * lo, model_ema.decay
        hip.check(self.lib.icamd_optim_ema(self.kind, m.param_arena.data_ptr() + 4 * lo, m.grad_arena.data_ptr() + 4 * lo,
                                           self.exp_avg.data_ptr() + 4 * lo,
                                           None if self.exp_avg_sq is None else self.exp_avg_sq.data_ptr() + 4 * lo, ema_ptr,
                                           m.shadow.data_ptr() + 2 * lo, hi - lo, float(g["lr"]), float(g["weight_decay"]),
                                           float(b1), float(b2), float(eps), self.step_count, float(grad_scale),
                                           float(decay), self.norm_clip.data_ptr() if use_clip else None,
                                           None if finite_flag is None else finite_flag.data_ptr(),
                                           self.skipped.data_ptr(), int(flags), hip.stream_ptr()), "optim_ema")

    def state_dict(self):
        sd = super().state_dict() if self.exp_avg_sq is not None else {
            "state": {"step": self.steps_taken, "exp_avg": self.exp_avg.cpu()},
            "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}
        sd["kind"] = self.kind
        return sd

    def load_state_dict(self, sd):
        st = sd["state"]
        self._load_step(st["step"])
        self.exp_avg.copy_(st["exp_avg"])
        if self.exp_avg_sq is not None:
            self.exp_avg_sq.copy_(st["exp_avg_sq"])
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update(src)


_FUSED_KINDS = {"adamw": OPT_ADAMW, "adam": OPT_ADAM, "sgd": OPT_SGD_NESTEROV, "nesterov": OPT_SGD_NESTEROV,
                "momentum": OPT_SGD_MOMENTUM, "lion": OPT_LION}


def create_optimizer(opt, lr, weight_decay, model, filter_bias_and_bn=True):
    """Name handling of reference optim_factory.py:50-122: lower-case, the last `_`-separated token selects the optimizer.
    `adamw` (the default recipe, train.py:50) is the fused hot path; sgd/nesterov, momentum, adam and lion are the other
    one-pass forms.  The remaining names need timm/apex classes (nadam, radam, adamp, sgdp, adadelta,
    adafactor, adahessian, rmsprop(tf), novograd, fused*, lookahead_*) and are not part of this build."""
    parts = opt.lower().split("_")
    name = parts[-1]
    if name not in _FUSED_KINDS or (len(parts) > 1 and parts[0] == "lookahead"):
        raise ValueError(f"optimizer '{opt}' is not built for the MI355X path (available: {sorted(_FUSED_KINDS)}); "
                         "see DESIGN.md")
    kind = _FUSED_KINDS[name]
    if kind == OPT_ADAMW:
        return FusedAdamW(model, lr=lr, weight_decay=weight_decay)
    if kind == OPT_LION:
        return FusedOptimizer(model, kind, lr=1e-4, weight_decay=weight_decay, betas=(0.9, 0.999))
    return FusedOptimizer(model, kind, lr=lr, weight_decay=weight_decay)
