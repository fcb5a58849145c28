"""GPU box tool: run one convolution layer shape through chosen entry points N times (for rocprofv3 --kernel-trace /
--pmc runs).  usage: python tools/one_layer.py Cin Cout k stride H [reps] [ops=fwd,dgrad,wgrad] [batch]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from imageclassification_amd import hip

cin, cout, k, st, h = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 64, 1, 1, 56))]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ops = (sys.argv[7] if len(sys.argv) > 7 else "fwd,dgrad,wgrad").split(",")
N = int(sys.argv[8]) if len(sys.argv) > 8 else 256
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
for _ in range(reps):
    if "fwd" in ops:
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, stats.data_ptr(), s))
    if "dgrad" in ops and cin % 64 == 0:
        hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, s))
    if "wgrad" in ops:
        hip.check(lib.icamd_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), wsb, s))
torch.cuda.synchronize()
