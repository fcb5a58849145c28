"""Per-layer timing of the three convolution kernels on the ResNet-50 (bs 256) shapes (GPU box tool).
usage: python tools/bench_layers.py [batch] [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageclassification_amd import hip

SHAPES = [  # Cin, Cout, k, stride, Hin, count
    (8, 64, 7, 2, 224, 1), (64, 64, 1, 1, 56, 1), (64, 64, 3, 1, 56, 3), (64, 256, 1, 1, 56, 4), (256, 64, 1, 1, 56, 2),
    (256, 128, 1, 1, 56, 1), (128, 128, 3, 2, 56, 1), (256, 512, 1, 2, 56, 1), (128, 512, 1, 1, 28, 4),
    (512, 128, 1, 1, 28, 3), (128, 128, 3, 1, 28, 3), (512, 256, 1, 1, 28, 1), (256, 256, 3, 2, 28, 1),
    (512, 1024, 1, 2, 28, 1), (256, 1024, 1, 1, 14, 6), (1024, 256, 1, 1, 14, 5), (256, 256, 3, 1, 14, 5),
    (1024, 512, 1, 1, 14, 1), (512, 512, 3, 2, 14, 1), (1024, 2048, 1, 2, 14, 1), (512, 2048, 1, 1, 7, 3),
    (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2),
]

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    lib = hip.load()
    s = hip.stream_ptr()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'shape':34s} {'M':>8s} {'GF':>7s} | {'fwd us':>8s} {'TF':>6s} {'GB/s':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    for (cin, cout, k, st, h, cnt) in SHAPES:
        pad = {1: 0, 3: 1, 7: 3}[k]
        d = hip.conv_desc(N, h, h, cin, cout, k, k, st, pad)
        x = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).to(torch.bfloat16)
        wt = w.permute(3, 1, 2, 0).contiguous()
        y = torch.empty(N, d.OH, d.OW, cout, dtype=torch.bfloat16, device="cuda")
        dy = torch.randn(N, d.OH, d.OW, cout, device="cuda").to(torch.bfloat16)
        dx = torch.empty_like(x)
        dw = torch.empty(cout, k, k, cin, device="cuda")
        stats = torch.empty(lib.icamd_conv2d_stats_rows(ctypes.byref(d)) * 2 * cout, device="cuda")
        wsb = lib.icamd_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        M = N * d.OH * d.OW
        cin_true = 3 if cin == 8 else cin
        gf = 2.0 * M * cout * cin_true * k * k / 1e9

        def timeit(fn):
            fn(); fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / reps

        if cin == 8:   # the ResNet stem runs on its own layout / entry points ([N][H][W+8][4] image, [Cout][8][8][4] filters)
            x4 = torch.randn(N, h, h + 8, 4, device="cuda").to(torch.bfloat16)
            w4 = (torch.randn(cout, 8, 8, 4, device="cuda") * 0.05).to(torch.bfloat16)
            wsb4 = lib.icamd_stem7x7s2_wgrad_workspace_bytes(N, h, h, cout)
            ws4 = torch.empty(wsb4, dtype=torch.uint8, device="cuda")
            dw4 = torch.empty(cout, 8, 8, 4, device="cuda")
            t_f = timeit(lambda: hip.check(lib.icamd_stem7x7s2_fwd(x4.data_ptr(), w4.data_ptr(), y.data_ptr(), None, stats.data_ptr(), 0, N, h, h, cout, s)))
            t_w = timeit(lambda: hip.check(lib.icamd_stem7x7s2_wgrad(x4.data_ptr(), dy.data_ptr(), dw4.data_ptr(), 0, ws4.data_ptr(), wsb4, N, h, h, cout, s)))
            gf_ = gf
            print(f"{'stem 7x7 s2 224 (rgb4 layout)':34s} {M:8d} {gf:7.1f} | {t_f:8.1f} {gf/t_f*1e3:6.0f} {0:6.0f} | {float('nan'):8.1f} {0:6.0f} fused {float('nan'):8.1f} | {t_w:8.1f} {gf/t_w*1e3:6.0f}")
            tot["fwd"] += t_f * cnt; totf["fwd"] += gf * cnt
            tot["wgrad"] += t_w * cnt; totf["wgrad"] += gf * cnt
            continue
        t_f = timeit(lambda: hip.check(lib.icamd_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, stats.data_ptr(), s)))
        t_d = float("nan")
        if cin != 8:
            t_d = timeit(lambda: hip.check(lib.icamd_conv2d_dgrad(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, s)))
        t_df = float("nan")
        if cin != 8:
            ybn = torch.randn(N, h, h, cin, device="cuda").to(torch.bfloat16)
            coef = torch.rand(4, cin, device="cuda") + 0.5
            rows = lib.icamd_conv2d_dgrad_stats_rows(ctypes.byref(d))
            part = torch.empty(rows * 2 * cin, device="cuda")
            use_mask = os.environ.get("FUSE_MASK", "0") == "1"
            f = hip.BnBwdFuse(ybn.data_ptr(), x.data_ptr() if use_mask else None, coef[0].data_ptr(), coef[1].data_ptr(),
                              coef[2].data_ptr(), coef[3].data_ptr(), part.data_ptr(), 1)
            t_df = timeit(lambda: hip.check(lib.icamd_conv2d_dgrad_bnbwd(ctypes.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, ctypes.byref(f), s)))
        t_w = timeit(lambda: hip.check(lib.icamd_conv2d_wgrad(ctypes.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), wsb, s)))
        io_gb = (x.numel() + y.numel()) * 2 / 1e9
        name = f"{cin}->{cout} k{k} s{st} {h}x{h} x{cnt}"
        print(f"{name:34s} {M:8d} {gf:7.1f} | {t_f:8.1f} {gf/t_f*1e3:6.0f} {io_gb/t_f*1e6:6.0f} | {t_d:8.1f} {gf/t_d*1e3:6.0f} fused {t_df:8.1f} | {t_w:8.1f} {gf/t_w*1e3:6.0f}")
        tot["fwd"] += t_f * cnt; totf["fwd"] += gf * cnt
        if cin != 8:
            tot["dgrad"] += t_d * cnt; totf["dgrad"] += gf * cnt
        tot["wgrad"] += t_w * cnt; totf["wgrad"] += gf * cnt
    for k in tot:
        print(f"total {k}: {tot[k]/1e3:.2f} ms, {totf[k]/tot[k]*1e3:.0f} TFLOP/s")

main()
