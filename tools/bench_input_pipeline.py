"""GPU box tool: throughput of the input pipeline on a generated JPEG folder -- the host path of datasets.TrainTransform
(PIL bicubic resize + numpy jitter / normalise / erase, what a DataLoader worker does per sample) next to the GPU pipeline
(host: JPEG decode only; device: icamd_image_pipeline).  usage: python tools/bench_input_pipeline.py [n_images] [batch]"""
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from imageclassification_amd.datasets import TrainTransform
from imageclassification_amd.gpu_pipeline import GpuImagePipeline


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rng = np.random.RandomState(0)
    blobs = []
    for i in range(n):      # photo-like sizes, smooth content + noise so that JPEG decoding costs what photos cost
        h, w = int(rng.randint(300, 520)), int(rng.randint(300, 520))
        yy, xx = np.mgrid[0:h, 0:w]
        a = (np.stack([yy * 255 // h, xx * 255 // w, (yy + xx) % 256], -1) + rng.randint(0, 40, (h, w, 3))).clip(0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format="JPEG", quality=90)
        blobs.append(buf.getvalue())
    t0 = time.perf_counter()
    decoded = [np.asarray(Image.open(io.BytesIO(b)).convert("RGB"), dtype=np.uint8) for b in blobs]
    t_dec = time.perf_counter() - t0
    tt = TrainTransform(224, 0.3, 0.25)
    t0 = time.perf_counter()
    for b in blobs:
        tt(Image.open(io.BytesIO(b)).convert("RGB"))
    t_host = time.perf_counter() - t0
    pipe = GpuImagePipeline(224, True)
    pipe(decoded[:B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 0
    for s in range(0, n - B + 1, B):
        pipe(decoded[s:s + B])
        reps += 1
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from imageclassification_amd import hip
    src, ddev = pipe._last[3], pipe._last[4]
    Bk, max_crop, kmax = pipe._last[:3]
    out = torch.empty(Bk, 3, 224, 224, device="cuda")
    a.record()
    for _ in range(10):
        hip.check(pipe.lib.icamd_image_pipeline(src.data_ptr(), ddev.data_ptr(), Bk, max_crop, 224, 224, 1, kmax, pipe.mean,
                                                pipe.std, out.data_ptr(), pipe._ws.data_ptr(), pipe._ws.numel(), hip.stream_ptr()))
    b.record()
    torch.cuda.synchronize()
    t_kern = a.elapsed_time(b) / 10 / 1e3
    print(f"{n} JPEGs (300-520 px), batch {B}:")
    print(f"  host JPEG decode only          : {n / t_dec:8.1f} img/s per core")
    print(f"  host decode + TrainTransform   : {n / t_host:8.1f} img/s per core   (the reference's per-worker path)")
    print(f"  GPU pipeline, host staging incl.: {reps * B / t_gpu:8.1f} img/s (one host thread packs + uploads; decode excluded)")
    print(f"  GPU pipeline kernels alone     : {Bk / t_kern:8.1f} img/s ({1e3 * t_kern:.2f} ms per batch of {Bk})")


main()
