"""Generates tests/golden/loss_curve_resnet50.json: the CPU oracle's own 24-step loss curve of ResNet-50 (fp32 accumulation, bf16
rounding points; oracle/engine_ref.py + oracle/resnet_ref.py) and the curve of its fp64 twin (the drift yardstick), ~6 minutes of
host work on 8 cores.  Until round 3 tests/test_model_gpu.py computed both inside the GPU test: 364 s of the driver's 900 s limit,
spent on the GPU box's host.  The curves are data (24 + 24 numbers + provenance); the GPU test replays the same seeded batches
through the HIP engine and compares; tests/test_oracle_cpu.py re-runs the first steps with the live oracle and pins the file to it.
Test infrastructure; no GPU, no product code.
Usage: python tests/golden/make_loss_curve_fixture.py [out.json] [threads]"""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import engine_ref as E  # noqa: E402
from oracle.resnet_ref import ResNetRef  # noqa: E402

C, B, HW, STEPS, NB = 10, 32, 128, 24, 4


def reference_model(seed=0):
    """timm-default ResNet-50 (zero-initialised last BatchNorm weight per block), the construction of test_model_gpu._timm_default_pair"""
    torch.manual_seed(seed)
    return ResNetRef("resnet50", C, bf16_points=True, zero_init_last=True)


def batches():
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(B, 3, HW, HW, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(NB)]
    return [data[i % NB] for i in range(STEPS)]


def schedules():
    return [1e-3 * i / 24 for i in range(STEPS)], [5e-4] * STEPS


class _As64(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x):
        return self.m(x.double())


def oracle_run(model, params, loader):
    lr, wd = schedules()
    opt = torch.optim.AdamW([{"params": list(params), "weight_decay": 5e-4}], lr=1e-3, weight_decay=0.0)
    tr = []
    E.train_one_epoch_ref(model, E.LabelSmoothingCrossEntropyRef(0.1), [(x.clone(), t.clone()) for x, t in loader], opt,
                          lr_schedule_values=lr, wd_schedule_values=wd, num_training_steps_per_epoch=STEPS, num_classes=C, trace=tr)
    return [t["loss"] for t in tr]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_curve_resnet50.json")
    if len(sys.argv) > 2:
        torch.set_num_threads(int(sys.argv[2]))
    t0 = time.time()
    ref = reference_model()
    ref64 = copy.deepcopy(ref).double()
    loader = batches()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        l_ref = oracle_run(ref, ref.parameters(), loader)
        l_64 = oracle_run(_As64(ref64), ref64.parameters(), loader)
    tmp = out + ".tmp"
    with open(tmp, "w") as f:
        json.dump({"what": "ResNet-50 (timm default init, seed 0), 10 classes, 24 AdamW steps on four cycled 32-image 128x128 batches "
                           "(generator seed 5), label smoothing 0.1, lr 1e-3 * i / 24, wd 5e-4: per-step loss of the CPU oracle and of "
                           "its fp64 twin", "generator": "tests/golden/make_loss_curve_fixture.py",
                   "torch": torch.__version__, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                   "oracle": l_ref, "oracle_fp64": l_64, "seconds": time.time() - t0, "threads": torch.get_num_threads()}, f, indent=1)
    os.replace(tmp, out)


if __name__ == "__main__":
    main()
