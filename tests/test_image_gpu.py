"""GPU input pipeline (include/icamd.h icamd_image_pipeline) against the numpy oracle of Pillow's arithmetic
(oracle/image_ref.py, pinned bit for bit to Pillow by tests/test_oracle_cpu.py) -- and, where Pillow is installed on the box,
against Pillow directly.  Integer stages (resize, crop, flips, colour jitter) are compared EXACTLY; the normalised fp32 tensor
to 1e-6; the erased box by position (exact) and by the statistics of its N(0,1) fill."""
import random

import numpy as np
import pytest
import torch

from oracle import image_ref as I

pytestmark = pytest.mark.gpu
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _images(seed, sizes):
    rng = np.random.RandomState(seed)
    out = []
    for k, (h, w) in enumerate(sizes):
        if k % 2:                                  # smooth content (gradients) next to noise: both exercise the rounding
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([yy * 255 // max(h - 1, 1), xx * 255 // max(w - 1, 1), (yy * 3 + xx * 5) % 256], -1).astype(np.uint8)
        else:
            a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        out.append(a)
    return out


def _oracle_train(a, pr, size):
    x = I.resize_u8(I.center_square(a), size, size, "bicubic")
    if pr["hflip"]:
        x = x[:, ::-1]
    if pr["vflip"]:
        x = x[::-1]
    x = np.ascontiguousarray(x)
    if pr["order"][0] >= 0:
        x = I.color_jitter(x, pr["order"], pr["factors"])
    return x


def test_train_pipeline_is_pillow_exact():
    from imageclassification_amd.gpu_pipeline import GpuImagePipeline
    size = 64
    sizes = [(37, 53), (120, 90), (20, 20), (64, 64), (200, 64), (65, 131), (64, 80), (300, 301)]
    imgs = _images(3, sizes)
    rnd = random.Random(5)
    params = []
    for k in range(len(imgs)):
        ops = [0, 1, 2]
        rnd.shuffle(ops)
        params.append({"hflip": k & 1, "vflip": (k >> 1) & 1, "order": tuple(ops) if k != 3 else (-1, -1, -1),
                       "factors": (rnd.uniform(0.7, 1.3), rnd.uniform(0.7, 1.3), rnd.uniform(0.7, 1.3)),
                       "erase": (5, 9, 20, 17) if k in (1, 6) else (0, 0, 0, 0), "seed": 1234 + k})
    pipe = GpuImagePipeline(size, True)
    out = pipe(imgs, params)
    torch.cuda.synchronize()
    u8 = pipe.last_uint8().cpu().numpy()
    for k, (a, pr) in enumerate(zip(imgs, params)):
        ref = _oracle_train(a, pr, size)
        assert np.array_equal(u8[k], ref), (k, sizes[k], int(np.abs(u8[k].astype(int) - ref.astype(int)).max()))
        t = I.to_tensor_normalize(ref, MEAN, STD)
        got = out[k].cpu().numpy()
        top, left, eh, ew = pr["erase"]
        if eh:
            box = got[:, top:top + eh, left:left + ew]
            assert abs(float(box.mean())) < 0.15 and 0.85 < float(box.std()) < 1.15           # N(0,1) fill, 1020 samples
            assert not np.array_equal(box[0], box[1])                                          # per-pixel, per-channel noise
            got = got.copy()
            got[:, top:top + eh, left:left + ew] = t[:, top:top + eh, left:left + ew]
        assert np.allclose(got, t, atol=1e-6, rtol=0), k
    # the same call again: identical, including the noise (seeded per image)
    out2 = pipe(imgs, params)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_eval_pipeline_and_pillow_direct():
    from imageclassification_amd.gpu_pipeline import GpuImagePipeline
    size = 48
    imgs = _images(7, [(60, 45), (48, 48), (31, 97), (224, 160)])
    pipe = GpuImagePipeline(size, False)
    out = pipe(imgs)
    torch.cuda.synchronize()
    u8 = pipe.last_uint8().cpu().numpy()
    try:
        from PIL import Image
    except ImportError:
        Image = None
    for k, a in enumerate(imgs):
        ref = I.resize_u8(a, size, size, "bilinear")           # Resize([s, s]): the whole image, squashed, bilinear
        assert np.array_equal(u8[k], ref), k
        if Image is not None:
            assert np.array_equal(u8[k], np.asarray(Image.fromarray(a).resize((size, size), Image.BILINEAR))), k
        assert np.allclose(out[k].cpu().numpy(), I.to_tensor_normalize(ref, MEAN, STD), atol=1e-6, rtol=0)


def test_full_size_batch_and_draws():
    """224x224 output from photo-sized inputs (the shape of the reference's data), parameters drawn as the host transform
    draws them; compared with the oracle on a few images of the batch."""
    from imageclassification_amd.gpu_pipeline import GpuImagePipeline, draw_train_params
    rng = np.random.RandomState(11)
    sizes = [(int(rng.randint(180, 520)), int(rng.randint(180, 520))) for _ in range(24)]
    imgs = _images(12, sizes)
    random.seed(99)
    params = [draw_train_params(224, 0.3, 0.25) for _ in imgs]
    assert any(p["erase"][2] > 0 for p in params) and any(p["hflip"] for p in params)
    pipe = GpuImagePipeline(224, True)
    out = pipe(imgs, params)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (24, 3, 224, 224) and torch.isfinite(out).all()
    u8 = pipe.last_uint8().cpu().numpy()
    for k in (0, 7, 23):
        assert np.array_equal(u8[k], _oracle_train(imgs[k], params[k], 224)), k
