"""Time icamd_conv1x1_bn_bwd_fused against the three launches it replaces (icamd_bn_bwd_from_gy_partials + icamd_conv2d_dgrad +
icamd_conv2d_wgrad) at ResNet-50's conv3 shapes, batch 256.  Usage: fused_bwd_probe.py [reps]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip
lib = hip.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = hip.stream_ptr()
for (N, H, Cin, Cout) in [(256, 56, 64, 256), (256, 28, 128, 512)]:
    d = hip.conv_desc(N, H, H, Cin, Cout, 1, 1, 1, 0)
    M = N * H * H
    g = (torch.randn(M, Cout, device="cuda") * (torch.rand(M, Cout, device="cuda") > 0.5)).bfloat16()
    y = (torch.randn(M, Cout, device="cuda") * 1.5 + 0.3).bfloat16()
    x = torch.randn(M, Cin, device="cuda").clamp_min(0).bfloat16()
    wt = (torch.randn(Cin, Cout, device="cuda") * Cin ** -0.5).bfloat16()
    mean, var = y.float().mean(0), y.float().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (0.5 + torch.rand(Cout, device="cuda")) * invstd
    rows = (M + 127) // 128
    part = torch.randn(rows, 2, Cout, device="cuda")
    dgam, dbet = torch.zeros(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    dx = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
    dy = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(Cout, Cin, device="cuda")
    bwsb = lib.icamd_bn_bwd_apply_workspace_bytes(Cout); bws = torch.zeros(bwsb, dtype=torch.uint8, device="cuda")
    fwsb = lib.icamd_conv1x1_bn_bwd_fused_workspace_bytes(ctypes.byref(d)); fws = torch.empty(max(fwsb, 16), dtype=torch.uint8, device="cuda")
    wsb = lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    P = hip.ptr

    def fused():
        hip.check(lib.icamd_conv1x1_bn_bwd_fused(ctypes.byref(d), P(part), rows, P(g), P(y), P(mean), P(invstd), P(scale), P(dgam), P(dbet),
                                                 P(x), P(wt), P(dx), P(dw), 0, P(bws), bwsb, P(fws), fwsb, s))

    def bn():
        hip.check(lib.icamd_bn_bwd_from_gy_partials(P(part), rows, P(g), P(y), P(mean), P(invstd), P(scale), P(dgam), P(dbet), P(dy), M, Cout, 0,
                                                    P(bws), bwsb, s))

    def dg():
        hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), P(dy), P(wt), P(dx), None, None, s))

    def wg():
        hip.check(lib.icamd_conv2d_wgrad(ctypes.byref(d), P(x), P(dy), P(dw), 0, P(ws), wsb, s))

    def timeit(fn):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    tb, td, tw = timeit(bn), timeit(dg), timeit(wg)
    tf = timeit(fused) if lib.icamd_conv1x1_bn_bwd_fused_supported(ctypes.byref(d)) else float("nan")
    gb = (2 * 2 * M * Cout + 2 * 2 * M * Cin) / 1e9
    print("%d->%d at %dx%d batch %d: bn apply %.1f + dgrad %.1f + wgrad %.1f = %.1f us;  fused %.1f us (%.2f GB -> %.0f GB/s)" %
          (Cin, Cout, H, H, N, tb, td, tw, tb + td + tw, tf, gb, gb / tf * 1e6))
