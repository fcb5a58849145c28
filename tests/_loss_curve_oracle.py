"""CPU side of tests/test_zz_loss_curve_gpu.py: the oracle's own 24-step loss curve (fp32 accumulation, bf16 rounding points)
and its fp64 twin, ~5 minutes of host work.  Run as a child process that tests/conftest.py starts when the GPU session is
collected, so that it overlaps with the rest of the suite instead of holding the GPU box for half of the driver's time limit
(round 3: 364 s of a 709 s suite).  Test infrastructure; no GPU, no product code.
Usage: python tests/_loss_curve_oracle.py <out.json> [threads]"""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    out = sys.argv[1]
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
        json.dump({"oracle": l_ref, "oracle_fp64": l_64, "seconds": time.time() - t0, "threads": torch.get_num_threads()}, f)
    os.replace(tmp, out)


if __name__ == "__main__":
    main()
