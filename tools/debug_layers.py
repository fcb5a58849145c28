"""Debug helper (GPU box): per-stage forward and backward comparison of the HIP ResNet against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ops_ref as R
from oracle.resnet_ref import ResNetRef
from tests.test_model_gpu import _pair, _xent_backward

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 64
C = 10
ref, net = _pair(arch, C)
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 3, HW, HW, generator=g)
y = torch.randint(0, C, (B,), generator=g)
acts = {}
def hook(name):
    def f(m, i, o):
        acts[name] = o.detach()
        o.retain_grad() if o.requires_grad else None
        acts[name + "_t"] = o
    return f
for li in range(1, 5):
    for bi, blk in enumerate(getattr(ref, f"layer{li}")):
        blk.register_forward_hook(hook(f"layer{li}.{bi}"))
ref.train()
out = ref(x)
loss = torch.nn.functional.cross_entropy(out, y, label_smoothing=0.1)
loss.backward()
net.train()
ws = net.pack(x.cuda())
logits = net.forward_packed(ws)
torch.cuda.synchronize()
def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)
print("x8", R.rel_l2(nchw(ws["x8"])[:, :3], R.bf16_round(x)))
import torch.nn.functional as F
with torch.no_grad():
    y0 = R.bf16_round(F.conv2d(R.bf16_round(x), R.bf16_round(ref.conv1.weight), None, 2, 3))
print("y0", R.rel_l2(nchw(ws["y0"]), y0))
for blk, b in zip(net.blocks, ws["blocks"]):
    print(blk["name"], "out", R.rel_l2(nchw(b["a"][-1]), acts[blk["name"]]))
print("logits", R.rel_l2(logits[:, :C].float().cpu(), out.detach()))
hl = _xent_backward(net, ws, y.cuda(), C, 0.1)
print("loss", hl, float(loss))
for name, p in ref.named_parameters():
    print(f"grad {name:40s} {R.rel_l2(net.grad_of(name), p.grad):.3e}")
