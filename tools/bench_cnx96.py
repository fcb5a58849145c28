"""GPU box: ConvNeXt-T's dim-96 Linear layers at batch 256 (M = 802 816): 96 -> 384 forward with GELU, 384 -> 96 data gradient
with GELU backward."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip
lib = hip.load(); s = hip.stream_ptr()
N, H = 256, 56
def timeit(fn, reps=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps
d1 = hip.conv_desc(N, H, H, 96, 384, 1, 1, 1, 0)
x = torch.randn(N, H, H, 96, device="cuda").bfloat16(); w1 = (torch.randn(384, 96, device="cuda") * 0.1).bfloat16(); b1 = torch.randn(384, device="cuda")
z = torch.empty(N, H, H, 384, dtype=torch.bfloat16, device="cuda"); a = torch.empty_like(z)
t1 = timeit(lambda: hip.check(lib.icamd_conv2d_fwd_gelu(ctypes.byref(d1), x.data_ptr(), w1.data_ptr(), z.data_ptr(), a.data_ptr(), b1.data_ptr(), s)))
t1b = timeit(lambda: hip.check(lib.icamd_conv2d_fwd_gelu(ctypes.byref(d1), x.data_ptr(), w1.data_ptr(), None, a.data_ptr(), b1.data_ptr(), s)))
d2 = hip.conv_desc(N, H, H, 384, 96, 1, 1, 1, 0)
dy = torch.randn(N, H, H, 96, device="cuda").bfloat16(); w2t = (torch.randn(384, 96, device="cuda") * 0.1).bfloat16()
dz = torch.empty_like(z)
t2 = timeit(lambda: hip.check(lib.icamd_conv2d_dgrad_gelu(ctypes.byref(d2), dy.data_ptr(), w2t.data_ptr(), z.data_ptr(), dz.data_ptr(), s)))
print(f"96->384 fwd + gelu (z and a): {t1:.1f} us ({(154+616*2)/t1*1e3/1e3:.2f} TB/s); a only: {t1b:.1f} us; 384->96 dgrad + gelu': {t2:.1f} us ({(154+616*2)/t2*1e3/1e3:.2f} TB/s)")
