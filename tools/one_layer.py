"""GPU box tool: run plain and fused data-gradient of one layer shape N times (for rocprofv3 --pmc runs)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip
cin, cout, k, st, h = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (256, 64, 1, 1, 56))]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
N = 256
lib = hip.load(); s = hip.stream_ptr()
pad = {1: 0, 3: 1}[k]
d = hip.conv_desc(N, h, h, cin, cout, k, k, st, pad)
x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
wt = (torch.randn(cin, k, k, cout, device="cuda") * 0.05).to(torch.bfloat16)
dy = torch.randn(N, d.OH, d.OW, cout, device="cuda").to(torch.bfloat16)
dx = torch.empty_like(x)
ybn = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
coef = torch.rand(4, cin, device="cuda") + 0.5
rows = lib.icamd_conv2d_dgrad_stats_rows(ctypes.byref(d))
part = torch.empty(rows * 2 * cin, device="cuda")
f = hip.BnBwdFuse(ybn.data_ptr(), None, coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), 1)
for _ in range(reps):
    hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, s))
for _ in range(reps):
    hip.check(lib.icamd_conv2d_dgrad_bnbwd(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, ctypes.byref(f), s))
torch.cuda.synchronize()
