"""GPU box tool: standalone bandwidth of the BatchNorm streaming kernels on one ResNet-50 activation shape.
usage: python tools/bench_bn.py [N H W C]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip

N, H, W, C = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (256, 56, 56, 256)
lib = hip.load(); s = hip.stream_ptr()
n = N * H * W * C
y = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
res = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
out = torch.empty_like(y); dy = torch.empty_like(y); dout = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
mask = torch.empty(n // 8, dtype=torch.uint8, device="cuda")
sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda") * 0.1
mean = torch.randn(C, device="cuda") * 0.1; invstd = torch.rand(C, device="cuda") + 0.5
dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
rows = N * H * W
wsb = lib.icamd_bn_bwd_workspace_bytes(rows, C); ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")

def timeit(fn, reps=20):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps

GB = n * 2 / 1e9
t = timeit(lambda: hip.check(lib.icamd_bn_apply(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, out.data_ptr(), None, n, C, 1, s)))
print(f"bn_apply (read y, write a): {t:.1f} us, {2 * GB / t * 1e3:.2f} TB/s")
t = timeit(lambda: hip.check(lib.icamd_bn_apply(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), out.data_ptr(), mask.data_ptr(), n, C, 1, s)))
print(f"bn_apply + residual + mask (2 reads, 1.06 writes): {t:.1f} us, {3.06 * GB / t * 1e3:.2f} TB/s")
t = timeit(lambda: hip.check(lib.icamd_bn_bwd(dout.data_ptr(), None, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), sc.data_ptr(), sh.data_ptr(), dg.data_ptr(), db.data_ptr(), dy.data_ptr(), None, None, rows, C, 1, 0, ws.data_ptr(), wsb, s)))
print(f"bn_bwd (4 reads, 1 write, 3 launches): {t:.1f} us, {5 * GB / t * 1e3:.2f} TB/s")
t = timeit(lambda: out.copy_(y))
print(f"torch copy_ (1 read, 1 write): {t:.1f} us, {2 * GB / t * 1e3:.2f} TB/s")
