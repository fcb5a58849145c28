"""GPU box: time the depthwise 7x7 kernels on ConvNeXt-T's four stage shapes at batch 256. usage: bench_dwconv.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip
lib = hip.load(); s = hip.stream_ptr()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def timeit(fn, reps=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
for (hw, C, cnt) in [(56, 96, 3), (28, 192, 3), (14, 384, 9), (7, 768, 3)]:
    x = torch.randn(N, hw, hw, C, device="cuda").bfloat16(); dy = torch.randn_like(x); ad = torch.randn_like(x)
    w = (torch.randn(7, 7, C, device="cuda") * 0.1).bfloat16(); b = torch.randn(C, device="cuda")
    y = torch.empty_like(x); dw = torch.empty(7, 7, C, device="cuda")
    wsb = lib.icamd_dwconv7_wgrad_workspace_bytes(N, hw, hw, C); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    tf = timeit(lambda: hip.check(lib.icamd_dwconv7_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, hw, hw, C, s)))
    td = timeit(lambda: hip.check(lib.icamd_dwconv7_dgrad(dy.data_ptr(), w.data_ptr(), ad.data_ptr(), y.data_ptr(), N, hw, hw, C, s)))
    tw = timeit(lambda: hip.check(lib.icamd_dwconv7_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), wsb, N, hw, hw, C, s)))
    mb = x.numel() * 2 / 1e6
    print(f"{hw}x{hw}x{C} x{cnt}: fwd {tf:7.1f} us ({2*mb/tf*1e3/1e3:5.2f} TB/s)  dgrad {td:7.1f} us  wgrad {tw:7.1f} us   [tensor {mb:.0f} MB; HBM floor fwd {2*mb/6.3e3*1e3/1e3*1e3:.0f} us]")
