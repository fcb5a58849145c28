"""Exponential moving average of the model (reference: timm.utils.ModelEmaV3 built at
/root/reference/train.py:198-201, updated at engine.py:68,77, evaluated at train.py:276,367).

timm semantics as recalled: an eval-mode deep copy; update() lerps EVERY floating state_dict entry
(parameters and BatchNorm running statistics) towards the model with weight 1-decay and copies integer
buffers; decay is the constant 0.9995 the reference passes (no warm-up).  Here the parameter lerp is fused
into the AdamW kernel (icamd_adamw_ema) and the buffer lerp is one icamd_lerp call."""
from . import hip


class ModelEmaV3:
    def __init__(self, model, decay=0.9999, device=None):
        self.decay = decay
        self.lib = hip.load()
        kw = {"img_size": model.img_size} if hasattr(model, "img_size") else {}
        self.module = type(model)(arch=model.arch, num_classes=model.num_classes, device=str(model.device), **kw)
        self.module.eval()
        self.set(model)

    @property
    def param_arena(self):
        return self.module.param_arena

    def set(self, model):
        self.module.param_arena.copy_(model.param_arena)
        self.module.buffer_arena.copy_(model.buffer_arena)
        self.module.num_batches_tracked = model.num_batches_tracked
        self.module.refresh_shadow()
        self.module._fold_dirty = True

    def after_fused_update(self, model, finite_flag=None):
        """Called by FusedAdamW.step after the fused parameter lerp: BN buffers + bookkeeping."""
        n = model.buffer_arena.numel()
        hip.check(self.lib.icamd_lerp(self.module.buffer_arena.data_ptr(), model.buffer_arena.data_ptr(), n,
                                      1.0 - self.decay, None if finite_flag is None else finite_flag.data_ptr(),
                                      hip.stream_ptr()), "ema buffers")
        self.module.num_batches_tracked = model.num_batches_tracked
        self.module.shadow_stale = True
        self.module._fold_dirty = True   # ResNet eval fast path re-folds BatchNorm before the next eval forward

    def update(self, model):
        """Stand-alone update (when the optimizer step was not the fused one)."""
        n = model.n_params
        hip.check(self.lib.icamd_lerp(self.module.param_arena.data_ptr(), model.param_arena.data_ptr(), n,
                                      1.0 - self.decay, None, hip.stream_ptr()), "ema params")
        self.after_fused_update(model)

    def prepare_eval(self):
        """bf16 filters of the EMA weights are only needed when the EMA model is evaluated."""
        self.module.refresh_shadow()
        self.module.shadow_stale = False

    def state_dict(self):
        return self.module.state_dict()
