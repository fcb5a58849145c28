"""Time icamd_bn_apply_conv1x1_fused against the two launches it replaces (icamd_bn_apply + icamd_conv2d_fwd with statistics) at
ResNet-50's block boundaries, batch 256.  Usage: fused_fwd_probe.py [reps]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip
lib = hip.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = hip.stream_ptr()
P = hip.ptr
for (N, H, K, Nout) in [(256, 56, 256, 64), (256, 56, 256, 128), (256, 28, 512, 128)]:
    d = hip.conv_desc(N, H, H, K, Nout, 1, 1, 1, 0)
    M = N * H * H
    y = (torch.randn(M, K, device="cuda") * 1.3).bfloat16()
    r = torch.randn(M, K, device="cuda").clamp_min(0).bfloat16()
    w = (torch.randn(Nout, K, device="cuda") * K ** -0.5).bfloat16()
    sc, sh = 0.5 + torch.rand(K, device="cuda"), torch.randn(K, device="cuda") * 0.3
    out = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    bits = torch.empty(M * K // 8, device="cuda", dtype=torch.uint8)
    y1 = torch.empty(M, Nout, device="cuda", dtype=torch.bfloat16)
    rows = lib.icamd_conv2d_stats_rows(ctypes.byref(d))
    stats = torch.empty(rows * 2 * Nout, device="cuda")

    def fused():
        hip.check(lib.icamd_bn_apply_conv1x1_fused(ctypes.byref(d), P(y), P(sc), P(sh), P(r), None, None, P(out), P(bits), P(w), P(y1), P(stats), s))

    def apply():
        hip.check(lib.icamd_bn_apply(P(y), P(sc), P(sh), P(r), P(out), P(bits), M * K, K, 1, s))

    def conv():
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), P(out), P(w), P(y1), None, None, P(stats), s))

    def timeit(fn):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    ta, tc = timeit(apply), timeit(conv)
    tf = timeit(fused) if lib.icamd_bn_apply_conv1x1_fused_supported(ctypes.byref(d)) else float("nan")
    gb = (3 * 2 * M * K + M * K / 8 + 2 * M * Nout) / 1e9
    print("%d->%d at %dx%d batch %d: bn apply %.1f + conv %.1f = %.1f us;  fused %.1f us (%.2f GB -> %.0f GB/s)" %
          (K, Nout, H, H, N, ta, tc, ta + tc, tf, gb, gb / tf * 1e6))
