"""GPU box: ConvNeXt-T's Linear (pointwise) layers of the four stages at batch 256, each entry point the model calls:
fc1 forward + GELU, fc2 forward (bias), fc2 data gradient + GELU', fc1 data gradient, both weight gradients (+ bias).
usage: python tools/bench_cnx_layers.py [batch] [reps] [stages=0,1,2,3]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip

lib = hip.load(); s = hip.stream_ptr()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
stages = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2,3").split(",")]


def timeit(fn):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


tot = 0.0
print(f"{'stage dim  M':22s} | {'fc1+gelu':>9s} {'fc2 fwd':>9s} {'fc2 dg+g':>9s} {'fc1 dgrad':>9s} {'fc1 wg':>9s} {'fc2 wg':>9s}  (us; TFLOP/s below)")
for si in stages:
    C, H, depth = [(96, 56, 3), (192, 28, 3), (384, 14, 9), (768, 7, 3)][si]
    M = N * H * H
    d1 = hip.conv_desc(N, H, H, C, 4 * C, 1, 1, 1, 0)
    d2 = hip.conv_desc(N, H, H, 4 * C, C, 1, 1, 1, 0)
    x = torch.randn(M, C, device="cuda").bfloat16()
    w1 = (torch.randn(4 * C, C, device="cuda") * 0.05).bfloat16(); w1t = w1.t().contiguous()
    w2 = (torch.randn(C, 4 * C, device="cuda") * 0.05).bfloat16(); w2t = w2.t().contiguous()
    b1 = torch.randn(4 * C, device="cuda"); b2 = torch.randn(C, device="cuda")
    z = torch.empty(M, 4 * C, dtype=torch.bfloat16, device="cuda"); a = torch.empty_like(z); dz = torch.empty_like(z)
    y = torch.empty(M, C, dtype=torch.bfloat16, device="cuda"); dy = torch.randn(M, C, device="cuda").bfloat16()
    dx = torch.empty_like(x)
    dw1 = torch.empty(4 * C, C, device="cuda"); dw2 = torch.empty(C, 4 * C, device="cuda")
    db1 = torch.empty(4 * C, device="cuda"); db2 = torch.empty(C, device="cuda")
    wsb = max(lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d1)), lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d2)))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    t = [
        timeit(lambda: hip.check(lib.icamd_conv2d_fwd_gelu(ctypes.byref(d1), x.data_ptr(), w1.data_ptr(), z.data_ptr(), a.data_ptr(), b1.data_ptr(), s))),
        timeit(lambda: hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d2), a.data_ptr(), w2.data_ptr(), y.data_ptr(), b2.data_ptr(), None, None, s))),
        timeit(lambda: hip.check(lib.icamd_conv2d_dgrad_gelu(ctypes.byref(d2), dy.data_ptr(), w2t.data_ptr(), z.data_ptr(), dz.data_ptr(), s))),
        timeit(lambda: hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d1), dz.data_ptr(), w1t.data_ptr(), dx.data_ptr(), None, None, s))),
        timeit(lambda: hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(d1), x.data_ptr(), dz.data_ptr(), dw1.data_ptr(), db1.data_ptr(), 0, ws.data_ptr(), wsb, s))),
        timeit(lambda: hip.check(lib.icamd_conv2d_wgrad_bias(ctypes.byref(d2), a.data_ptr(), dy.data_ptr(), dw2.data_ptr(), db2.data_ptr(), 0, ws.data_ptr(), wsb, s))),
    ]
    gf = 2.0 * M * C * 4 * C / 1e9
    print(f"{si} {C:4d} {M:8d} {gf:5.1f}GF | " + " ".join(f"{v:9.1f}" for v in t))
    print(f"{'':22s} | " + " ".join(f"{gf / v * 1e3:9.0f}" for v in t))
    # per training step under mixup: forward twice (the extra train-mode forward), backward once
    step = depth * (2 * (t[0] + t[1]) + t[2] + t[3] + t[4] + t[5])
    tot += step
    print(f"{'':22s} | per step x{depth} blocks (fwd twice): {step / 1e3:.2f} ms")
print(f"total pointwise per step: {tot / 1e3:.2f} ms")
