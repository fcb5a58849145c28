"""Time one pointwise weight gradient (icamd_conv2d_wgrad, kernel + slab reduce) at a Linear-layer shape and check it against
torch.  Usage: wgrad_probe.py M Cin Cout [reps]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip
lib = hip.load()
M, Cin, Cout = [int(a) for a in sys.argv[1:4]]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
d = hip.conv_desc(1, M, 1, Cin, Cout, 1, 1, 1, 0)
x = torch.randn(M, Cin, device="cuda").bfloat16(); dy = torch.randn(M, Cout, device="cuda").bfloat16()
dw = torch.empty(Cout, Cin, device="cuda")
wsb = lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
def run():
    rc = lib.icamd_conv2d_wgrad(ctypes.byref(d), hip.ptr(x), hip.ptr(dy), hip.ptr(dw), 0, hip.ptr(ws), wsb, hip.stream_ptr()); assert rc == 0
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
ref = dy[:, :256].float().t() @ x.float()
err = ((dw[:256] - ref).norm() / ref.norm()).item()
print("wgrad M %d Cin %d Cout %d: %.1f us  %.0f TFLOP/s  slab %.1f MB (rel err first 256 filters %.1e)" % (M, Cin, Cout, us, 2.0 * M * Cin * Cout / us * 1e-6, wsb / 1e6, err))
