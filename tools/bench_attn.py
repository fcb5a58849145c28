"""GPU box: the attention kernels at ViT-B/16's shape (batch 256, T = 197, 12 heads of 64), forward and backward, with a
correctness check against torch's fp32 softmax attention on a few images.  usage: python tools/bench_attn.py [batch] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip
lib = hip.load(); s = hip.stream_ptr()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, H, D = 197, (int(sys.argv[3]) if len(sys.argv) > 3 else 12), 64
scale = D ** -0.5
g = torch.Generator(device="cuda").manual_seed(3)
qkv = (torch.randn(B * T, 3 * H * D, device="cuda", generator=g)).bfloat16()
dout = (torch.randn(B * T, H * D, device="cuda", generator=g) * 0.1).bfloat16()
out = torch.empty(B * T, H * D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, T, device="cuda"); delta = torch.empty_like(lse)
dqkv = torch.empty_like(qkv)
def fwd(): hip.check(lib.icamd_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, T, H, D, scale, s))
def bwd(): hip.check(lib.icamd_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), B, T, H, D, scale, s))
def timeit(fn):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
tf, tb = timeit(fwd), timeit(bwd)
# reference on the first 4 images
nb = 4
x = qkv[: nb * T].float().reshape(nb, T, 3, H, D).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)   # [3][nb][H][T][D]
q, k, v = x[0], x[1], x[2]
p = torch.softmax(q @ k.transpose(-1, -2) * scale, -1)
o = (p @ v).permute(0, 2, 1, 3).reshape(nb * T, H * D)
o.backward(dout[: nb * T].float())
ref_d = x.grad.permute(1, 3, 0, 2, 4).reshape(nb * T, 3 * H * D)
e_o = ((out[: nb * T].float() - o).norm() / o.norm()).item()
e_d = ((dqkv[: nb * T].float() - ref_d).norm() / ref_d.norm()).item()
gf_f = 4.0 * B * H * T * T * D / 1e9
print(f"attention B {B} T {T} H {H}: fwd {tf:.1f} us ({gf_f / tf * 1e3:.0f} TFLOP/s), bwd {tb:.1f} us ({2.5 * gf_f / tb * 1e3:.0f} TFLOP/s on 5 products); "
      f"rel err out {e_o:.1e}, dqkv {e_d:.1e}")
