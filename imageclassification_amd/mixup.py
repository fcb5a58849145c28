"""Mixup / CutMix parameter sampling and the loss criteria objects (reference: timm.data.Mixup built at
/root/reference/train.py:172-185, called at engine.py:44; criteria chosen at train.py:256-261).

timm is not in the reference tree; the sampling below follows timm.data.mixup ("batch" mode) as recalled:
lambda ~ Beta(alpha, alpha) and the cut box come from NUMPY's global RNG (seeded by train.py:118), in this
call order: rand() < prob, rand() < switch_prob (only when both alphas > 0), beta(), then for cutmix
randint(cy), randint(cx).  The pixel mixing itself runs inside the input packing kernel (icamd_pack_input)
and the soft targets inside the loss kernel (icamd_softmax_xent), so nothing is materialised on the host.
"""
import numpy as np


def rand_bbox(img_shape, lam, margin=0.0):
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y)
    cx = np.random.randint(0 + margin_x, img_w - margin_x)
    yl = int(np.clip(cy - cut_h // 2, 0, img_h))
    yh = int(np.clip(cy + cut_h // 2, 0, img_h))
    xl = int(np.clip(cx - cut_w // 2, 0, img_w))
    xh = int(np.clip(cx + cut_w // 2, 0, img_w))
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax):
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]))
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]))
    yl = np.random.randint(0, img_h - cut_h)
    xl = np.random.randint(0, img_w - cut_w)
    return int(yl), int(yl + cut_h), int(xl), int(xl + cut_w)


class Mixup:
    """Holds the mixup configuration and draws per-batch parameters.

    `sample(shape)` -> (mode, lam, box): mode 0 = untouched (lam 1), 1 = mixup, 2 = cutmix (lam already
    corrected to the pasted area).  `label_smoothing` / `num_classes` feed the loss kernel's soft targets:
    lam*onehot_s(y) + (1-lam)*onehot_s(y.flip(0)), on = 1-s+s/C, off = s/C."""

    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        if mode != "batch":
            raise ValueError("only timm's 'batch' mixup mode (the reference default, train.py:79) is on the hot path")
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.cutmix_minmax = cutmix_minmax
        if cutmix_minmax is not None:
            assert len(cutmix_minmax) == 2
            self.cutmix_alpha = 1.0
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.correct_lam = correct_lam
        self.mixup_enabled = True

    def sample(self, shape):
        return sample_params(self, shape)


_MIXUP_FIELDS = ("mixup_alpha", "cutmix_alpha", "mix_prob", "switch_prob", "label_smoothing", "num_classes")


def sample_params(cfg, shape):
    """Per-batch (mode, lam, box) drawn from ANY object that carries timm.data.Mixup's public fields -- this package's Mixup
    or the object the reference builds at train.py:176-185 and passes as `mixup_fn` (engine.py:44): `mixup_alpha`,
    `cutmix_alpha`, `cutmix_minmax`, `mix_prob`, `switch_prob`, `correct_lam`, `mixup_enabled`, `mode`.  The fields are read
    at every call (timm's `mixup_off_epoch` flips `mixup_enabled` on the live object).  Only mode "batch" (the reference
    default, train.py:79) is built: "pair" / "elem" draw per-sample parameters the fused packing kernel does not take."""
    missing = [f for f in _MIXUP_FIELDS if not hasattr(cfg, f)]
    if missing:
        raise TypeError(f"mixup_fn of type {type(cfg).__name__} lacks timm.data.Mixup's field(s) {missing}: the MI355X "
                        "engine draws the mixing parameters itself from the object's configuration")
    mode = getattr(cfg, "mode", "batch")
    if mode != "batch":
        raise ValueError(f"mixup mode {mode!r} is not built for the MI355X path (only timm's 'batch' mode, the reference "
                         "default train.py:79)")
    assert shape[0] % 2 == 0, "Batch size should be even when using this"
    minmax = getattr(cfg, "cutmix_minmax", None)
    lam, use_cutmix = 1.0, False
    if getattr(cfg, "mixup_enabled", True) and np.random.rand() < cfg.mix_prob:
        if cfg.mixup_alpha > 0.0 and cfg.cutmix_alpha > 0.0:
            use_cutmix = np.random.rand() < cfg.switch_prob
            alpha = cfg.cutmix_alpha if use_cutmix else cfg.mixup_alpha
            lam = float(np.random.beta(alpha, alpha))
        elif cfg.mixup_alpha > 0.0:
            lam = float(np.random.beta(cfg.mixup_alpha, cfg.mixup_alpha))
        elif cfg.cutmix_alpha > 0.0:
            use_cutmix = True
            lam = float(np.random.beta(cfg.cutmix_alpha, cfg.cutmix_alpha))
        else:
            raise AssertionError("One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.")
    if lam == 1.0:
        return 0, 1.0, (0, 0, 0, 0)
    if use_cutmix:
        if minmax is not None:
            box = rand_bbox_minmax(shape, minmax)
        else:
            box = rand_bbox(shape, lam)
        if getattr(cfg, "correct_lam", True) or minmax is not None:
            area = (box[1] - box[0]) * (box[3] - box[2])
            lam = 1.0 - area / float(shape[-2] * shape[-1])
        return 2, lam, box
    return 1, lam, (0, 0, 0, 0)


class _Criterion:
    smoothing = 0.0
    soft = False

    def __repr__(self):
        return f"{type(self).__name__}()"


class CrossEntropyLoss(_Criterion):
    """torch.nn.CrossEntropyLoss (reference train.py:261; evaluate always uses it, engine.py:147)."""


class LabelSmoothingCrossEntropy(_Criterion):
    """timm.loss.LabelSmoothingCrossEntropy == F.cross_entropy(label_smoothing=s) (reference train.py:259)."""

    def __init__(self, smoothing=0.1):
        assert smoothing < 1.0
        self.smoothing = smoothing


class SoftTargetCrossEntropy(_Criterion):
    """timm.loss.SoftTargetCrossEntropy over Mixup's soft targets (reference train.py:257); the smoothing
    and lambda come from the Mixup object of the same step."""
    soft = True
