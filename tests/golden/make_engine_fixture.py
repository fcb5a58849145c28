"""Generates tests/golden/engine_trace.json by executing the REFERENCE's own loop code on CPU.

Run in the build container only (needs /root/reference; nothing here travels to the GPU box except the JSON):
    python tests/golden/make_engine_fixture.py

What runs: /root/reference/engine.py train_one_epoch + evaluate and /root/reference/utils.py cosine_scheduler,
imported unmodified.  Their imports of packages that are not installed (timm, tensorboardX) are satisfied by
in-memory stand-in modules exposing ONLY the names the reference imports; the arithmetic stand-ins
(Mixup / accuracy / ModelEmaV3) are the oracle's own restatements, so this fixture pins the LOOP (ordering,
skip rules, averaging, return keys, schedule injection) -- not timm.  torch.cuda.synchronize (engine.py:79) is
patched to a no-op because this container has no GPU.  Inputs are seeded torch/numpy draws recorded in the JSON.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import engine_ref as E  # noqa: E402


def install_standins():
    timm = types.ModuleType("timm")
    data = types.ModuleType("timm.data")
    utils_m = types.ModuleType("timm.utils")
    models = types.ModuleType("timm.models")
    loss = types.ModuleType("timm.loss")
    optim = types.ModuleType("timm.optim")
    data.Mixup = E.MixupRef
    utils_m.accuracy = E.accuracy_ref
    utils_m.ModelEmaV3 = E.ModelEmaRef
    utils_m.get_state_dict = lambda m, *a, **k: m.state_dict()
    models.create_model = None
    loss.LabelSmoothingCrossEntropy = E.LabelSmoothingCrossEntropyRef
    loss.SoftTargetCrossEntropy = E.SoftTargetCrossEntropyRef
    timm.data, timm.utils, timm.models, timm.loss, timm.optim = data, utils_m, models, loss, optim
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    for name, mod in (("timm", timm), ("timm.data", data), ("timm.utils", utils_m), ("timm.models", models),
                      ("timm.loss", loss), ("timm.optim", optim), ("tensorboardX", tbx)):
        sys.modules[name] = mod
    torch.cuda.synchronize = lambda *a, **k: None


class TinyNet(torch.nn.Module):
    """conv-BN-ReLU-pool-FC: has BatchNorm buffers (EMA / running-stat paths) and is cheap on CPU."""

    def __init__(self, num_classes):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 8, 3, padding=1, bias=False)
        self.bn = torch.nn.BatchNorm2d(8)
        self.fc = torch.nn.Linear(8, num_classes)

    def forward(self, x):
        x = torch.relu(self.bn(self.conv(x)))
        return self.fc(x.mean(dim=(2, 3)))


def make_data(seed, nbatch, B, C, hw=8):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 3, hw, hw, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(nbatch)]


def run_case(engine, utils, name, C, B, nbatch, update_freq, mixup, use_ema, nan_step=None):
    torch.manual_seed(1234)
    np.random.seed(1234)
    model = TinyNet(C)
    init = {k: v.clone() for k, v in model.state_dict().items()}
    data = make_data(77, nbatch, B, C)
    if nan_step is not None:
        data[nan_step][0][0, 0, 0, 0] = float("nan")
    steps = nbatch // update_freq
    lr = utils.cosine_scheduler(1e-2, 1e-5, 2, steps, warmup_epochs=1)
    wd = utils.cosine_scheduler(5e-2, 5e-3, 2, steps)
    opt = torch.optim.AdamW([{"params": list(model.parameters()), "weight_decay": 5e-2}], lr=1e-2, weight_decay=0.0)
    mixup_fn = E.MixupRef(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C) if mixup else None
    crit = E.SoftTargetCrossEntropyRef() if mixup else E.LabelSmoothingCrossEntropyRef(0.1)
    ema = E.ModelEmaRef(model, decay=0.9) if use_ema else None
    # the CPU aliasing quirk (SURVEY Appx C.2): .to('cpu') returns the same tensor, so hand the loop fresh clones
    loader = [(x.clone(), y.clone()) for x, y in data]
    train_stats = engine.train_one_epoch(model, crit, loader, opt, torch.device("cpu"), 0, None, None, ema, mixup_fn,
                                         start_steps=0, lr_schedule_values=lr, wd_schedule_values=wd,
                                         num_training_steps_per_epoch=steps, update_freq=update_freq, use_amp=False,
                                         num_classes=C)
    eval_stats = engine.evaluate(make_data(78, 2, B + B // 2, C), model, torch.device("cpu"), num_classes=C)
    return {"name": name, "C": C, "B": B, "nbatch": nbatch, "update_freq": update_freq, "mixup": mixup,
            "use_ema": use_ema, "nan_step": nan_step,
            "train_stats": train_stats, "eval_stats": eval_stats,
            "lr": [float(v) for v in lr], "wd": [float(v) for v in wd],
            "final_fc_bias": model.state_dict()["fc.bias"].tolist(),
            "final_bn_running_mean": model.state_dict()["bn.running_mean"].tolist(),
            "ema_fc_bias": ema.module.state_dict()["fc.bias"].tolist() if ema else None,
            "init_fc_bias": init["fc.bias"].tolist()}


def main():
    install_standins()
    sys.path.insert(0, REF)
    import engine  # the reference's engine.py, unmodified
    import utils   # the reference's utils.py, unmodified
    cases = [
        run_case(engine, utils, "plain_ls", C=3, B=4, nbatch=6, update_freq=1, mixup=False, use_ema=False),
        run_case(engine, utils, "accum2_ema", C=2, B=4, nbatch=8, update_freq=2, mixup=False, use_ema=True),
        run_case(engine, utils, "nan_skip", C=3, B=4, nbatch=5, update_freq=1, mixup=False, use_ema=True, nan_step=2),
        run_case(engine, utils, "mixup_cpu_alias", C=4, B=4, nbatch=4, update_freq=1, mixup=True, use_ema=False),
    ]
    sched = {"args": [[1e-3, 1e-6, 3, 7, 1], [5e-4, 5e-6, 2, 5, 0]],
             "values": [[float(v) for v in utils.cosine_scheduler(1e-3, 1e-6, 3, 7, warmup_epochs=1)],
                        [float(v) for v in utils.cosine_scheduler(5e-4, 5e-6, 2, 5)]]}
    out = {"generator": "tests/golden/make_engine_fixture.py", "torch": torch.__version__, "cases": cases,
           "cosine_scheduler": sched}
    with open(os.path.join(HERE, "engine_trace.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote engine_trace.json:", [(c["name"], c["train_stats"]) for c in cases])


if __name__ == "__main__":
    main()
