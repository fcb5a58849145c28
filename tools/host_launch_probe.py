"""Host cost of one C-ABI launch (round 4: bench.py's host_enqueue_ms is 14 ms of a 19.3 ms step, 29 us per launch).
Measures, with time.perf_counter around a burst of calls and NO device wait inside the burst: a tiny kernel on an idle GPU,
the same call with a kernel that keeps the GPU busy ~60 us (does the host block once many launches are outstanding?), the same
through a pre-bound ctypes argument tuple, and torch's own launch for comparison."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip  # noqa: E402

lib = hip.load()
dev = torch.device("cuda")
C = 256


def burst(numel, n, label, prebound=False):
    y = torch.randn(numel // C, C, device=dev).bfloat16()
    out = torch.empty_like(y)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    s = hip.stream_ptr()
    torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    if prebound:
        fn = lib.icamd_bn_apply
        args = (hip.ptr(y), hip.ptr(scale), hip.ptr(shift), None, hip.ptr(out), None, numel, C, 1, s)
        for i in range(n):
            fn(*args)
            if i % 250 == 249:
                marks.append(time.perf_counter())
    else:
        for i in range(n):
            lib.icamd_bn_apply(hip.ptr(y), hip.ptr(scale), hip.ptr(shift), None, hip.ptr(out), None, numel, C, 1, hip.stream_ptr())
            if i % 250 == 249:
                marks.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = [1e6 * (b - a) / 250 for a, b in zip([t0] + marks[:-1], marks)]
    print(f"{label:46s} host {1e6 * (t1 - t0) / n:7.2f} us/call   drain {1e3 * (t2 - t1):8.2f} ms   per 250 calls: "
          + " ".join(f"{p:.1f}" for p in per))


burst(C * 64, 3000, "tiny kernel, idle GPU")
burst(C * 64, 3000, "tiny kernel, idle GPU, pre-bound args", prebound=True)
burst(C * 200704 // 4, 3000, "13 MB kernel (~10 us)")
burst(C * 200704 * 2, 3000, "103 MB kernel (~45 us), pre-bound args", prebound=True)
burst(C * 200704 * 2, 6000, "103 MB kernel (~45 us), 6000 calls", prebound=True)
x = torch.randn(64, 256, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    x.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{'torch x.add_(1) tiny':46s} host {1e6 * (t1 - t0) / 3000:7.2f} us/call")
ev = [torch.cuda.Event() for _ in range(2000)]
side = torch.cuda.Stream()
t0 = time.perf_counter()
for e in ev:
    e.record()
    side.wait_event(e)
t1 = time.perf_counter()
print(f"{'event record + other-stream wait':46s} host {1e6 * (t1 - t0) / 2000:7.2f} us/pair")
