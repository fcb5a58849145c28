"""ConvNeXt-T block tail at batch 256: fc2 (bias) + icamd_layerscale_fwd against fc2 with the layer scale folded into its filter
(bias + residual addend in the GEMM's store pass); backward: icamd_layerscale_bwd + fc2 data gradient against the data gradient
alone.  Usage: ls_fuse_probe.py [reps]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclassification_amd import hip
lib = hip.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = hip.stream_ptr()
P = hip.ptr


def timeit(fn):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


tot = [0.0, 0.0, 0.0]
for (N, H, dim, nblk) in [(256, 56, 96, 3), (256, 28, 192, 3), (256, 14, 384, 9), (256, 7, 768, 3)]:
    d = hip.conv_desc(N, H, H, 4 * dim, dim, 1, 1, 1, 0)
    M = N * H * H
    a = torch.randn(M, 4 * dim, device="cuda").bfloat16()
    x = torch.randn(M, dim, device="cuda").bfloat16()
    w = (torch.randn(dim, 4 * dim, device="cuda") * (4 * dim) ** -0.5).bfloat16()
    b = torch.randn(dim, device="cuda") * 0.1
    gamma = 0.3 + torch.rand(dim, device="cuda")
    keep = ((torch.rand(N, device="cuda") < 0.95).float() / 0.95)
    z2 = torch.empty(M, dim, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, dim, device="cuda", dtype=torch.bfloat16)

    def fc2_plain():
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), P(a), P(w), P(z2), P(b), None, None, s))

    def fc2_addend():
        hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), P(a), P(w), P(out), P(b), P(x), None, s))

    def ls():
        hip.check(lib.icamd_layerscale_fwd(P(z2), P(x), P(gamma), P(keep), P(out), M, dim, H * H, s))

    t1, t2, t3 = timeit(fc2_plain), timeit(ls), timeit(fc2_addend)
    print("dim %d at %dx%d: fc2 %.1f + layer scale %.1f = %.1f us;  fc2 with addend %.1f us   (x%d blocks)" %
          (dim, H, H, t1, t2, t1 + t2, t3, nblk))
    tot[0] += nblk * (t1 + t2); tot[1] += nblk * t3; tot[2] += nblk * t2
print("per forward: %.2f ms -> %.2f ms (layer scale alone %.2f ms)" % (tot[0] / 1e3, tot[1] / 1e3, tot[2] / 1e3))
