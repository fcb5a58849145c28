"""Time one pointwise forward (icamd_conv2d_fwd without statistics) at a ViT Linear shape. Usage: gemm_probe.py M N K"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip
lib = hip.load()
M, N, K = [int(a) for a in sys.argv[1:4]]
d = hip.conv_desc(1, M, 1, K, N, 1, 1, 1, 0)
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def run():
    rc = lib.icamd_conv2d_fwd(ctypes.byref(d), hip.ptr(x), hip.ptr(w), hip.ptr(y), None, None, None, hip.stream_ptr()); assert rc == 0
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
ref = (x[:4096].float() @ w.float().t())
err = ((y[:4096].float() - ref).norm() / ref.norm()).item()
ref2 = (x[-4096:].float() @ w.float().t())
err2 = ((y[-4096:].float() - ref2).norm() / ref2.norm()).item()
# whole output against an fp32 product (the tail round's tiles lie wherever the tile order puts them), and run-to-run bits
full = 0.0
for r0 in range(0, M, 8192):
    ref = x[r0:r0 + 8192].float() @ w.float().t()
    full = max(full, ((y[r0:r0 + 8192].float() - ref).norm() / ref.norm()).item())
y1 = y.clone(); run(); torch.cuda.synchronize()
print("M %d N %d K %d: %.1f us  %.0f TFLOP/s  (rel err first / last 4096 rows %.1e %.1e, worst 8192-row block %.1e, bit-stable %s)" %
      (M, N, K, us, 2.0 * M * N * K / us * 1e-6, err, err2, full, bool(torch.equal(y, y1))))
