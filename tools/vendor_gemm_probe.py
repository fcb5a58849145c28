"""Comparison only (never used by the product): what the vendor GEMM behind torch.matmul (hipBLASLt / rocBLAS) reaches on the
ViT-B/16 Linear shapes on this board, next to tools/gemm_probe.py for gemm_nt.hip.  Usage: python tools/vendor_gemm_probe.py"""
import torch, time
for (M,N,K) in [(50432,3072,768),(50432,768,3072),(50432,2304,768),(50432,768,768)]:
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    for _ in range(5): c=a@b.t()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): c=a@b.t()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f"torch.matmul (hipBLASLt/rocBLAS) M {M} N {N} K {K}: {us:.1f} us {2.0*M*N*K/us*1e-6:.0f} TFLOP/s")
