"""GPU box tool: HIP-event time of one layer's entry points, each alone on the GPU.
usage: python tools/time_one_layer.py Cin Cout k stride H [reps] [batch]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from imageclassification_amd import hip

cin, cout, k, st, h = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
N = int(sys.argv[7]) if len(sys.argv) > 7 else 256
lib = hip.load()
s = hip.stream_ptr()
pad = {1: 0, 3: 1, 7: 3}[k]
d = hip.conv_desc(N, h, h, cin, cout, k, k, st, pad)
x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).to(torch.bfloat16)
wt = w.permute(3, 1, 2, 0).contiguous()
y = torch.empty(N, d.OH, d.OW, cout, dtype=torch.bfloat16, device="cuda")
dy = torch.randn(N, d.OH, d.OW, cout, device="cuda").to(torch.bfloat16)
dx = torch.empty_like(x)
dw = torch.empty(cout, k, k, cin, device="cuda")
stats = torch.empty(lib.icamd_conv2d_stats_rows(ctypes.byref(d)) * 2 * cout, device="cuda")
wsb = lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
ops = {"fwd": lambda: lib.icamd_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, stats.data_ptr(), s),
       "dgrad": lambda: lib.icamd_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, s),
       "wgrad": lambda: lib.icamd_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), wsb, s)}
flops = 2.0 * N * d.OH * d.OW * cout * cin * k * k
for name, fn in ops.items():
    for _ in range(3):
        hip.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.check(fn())
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print(f"{cin}->{cout} {k}x{k}/{st} at {h}^2 batch {N} {name:6s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")
