"""Eval-forward throughput of the HIP ResNet with and without the BatchNorm-folded fast path.
Usage: python tools/eval_probe.py [arch] [batch]"""
import sys, time, torch
sys.path.insert(0, ".")
from imageclassification_amd.nets import ResNet
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 384
net = ResNet(arch, 1000, seed=0).eval()
x = torch.randn(B, 3, 224, 224, device="cuda")
for fold in (True, False):
    net.fold_eval = fold
    for _ in range(3):
        net(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        net(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{arch} eval batch {B} fold={fold}: {dt * 1e3:.2f} ms  {B / dt:.0f} img/s")
