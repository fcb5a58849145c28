"""Input pipeline on the GPU: the reference's train / eval transforms after JPEG decoding
(/root/reference/datasets.py:121-144; loaders train.py:152-170) as ONE C-ABI call per batch (icamd_image_pipeline).

The host keeps what only it can do -- file I/O and JPEG decoding (PIL) -- and the random DECISIONS (flips, colour-jitter
order and factors, erase box), drawn per image from Python's `random` in the order the host transform of datasets.py
draws them; the pixels never take the per-sample CPU path of the reference (bicubic resize, enhancers, normalisation and
erasing run as HIP kernels with Pillow's exact integer arithmetic).  The result is the fp32 NCHW batch train_one_epoch /
evaluate expect, already on the device.
"""
import ctypes
import math
import random

import numpy as np
import torch

from . import hip

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def center_square_box(h, w):
    """timm RandomResizedCrop(scale=(1,1), ratio=(1,1)): whole image if square, else the centred min(W,H) square."""
    side = min(h, w)
    return (h - side) // 2, (w - side) // 2, side, side


def draw_train_params(size, color_jitter=0.3, reprob=0.25, hflip=0.5, vflip=0.5, rng=random):
    """One image's random decisions, in the draw order of datasets.TrainTransform (the host path of the same recipe)."""
    d = {"hflip": int(rng.random() < hflip), "vflip": int(rng.random() < vflip), "order": (-1, -1, -1),
         "factors": (1.0, 1.0, 1.0), "erase": (0, 0, 0, 0), "seed": 0}
    if color_jitter and color_jitter > 0:
        ops = [0, 1, 2]
        rng.shuffle(ops)
        f = [1.0, 1.0, 1.0]
        for op in ops:
            f[op] = rng.uniform(max(0.0, 1 - color_jitter), 1 + color_jitter)
        d["order"], d["factors"] = tuple(ops), tuple(f)
    if reprob > 0 and rng.random() < reprob:      # timm RandomErasing, mode 'pixel', one box, <= 10 attempts
        area = size * size
        for _ in range(10):
            target = rng.uniform(0.02, 1 / 3) * area
            aspect = math.exp(rng.uniform(math.log(0.3), math.log(1 / 0.3)))
            h, w = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
            if 0 < h < size and 0 < w < size:
                d["erase"] = (rng.randint(0, size - h), rng.randint(0, size - w), h, w)
                d["seed"] = rng.getrandbits(32)
                break
    return d


class GpuImagePipeline:
    """uint8 HWC numpy images (any sizes) -> fp32 [B, 3, size, size] on the device."""

    def __init__(self, size, train, color_jitter=0.3, reprob=0.25, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD,
                 device="cuda"):
        hip.require_gpu()
        self.lib = hip.load()
        self.size, self.train = int(size), bool(train)
        self.color_jitter, self.reprob = color_jitter, reprob
        self.filter = 1 if train else 0          # bicubic for training (datasets.py:131), torchvision's bilinear for eval
        self.mean = (ctypes.c_float * 3)(*mean)
        self.std = (ctypes.c_float * 3)(*std)
        self.device = torch.device(device)
        self._ws = None
        # two pinned staging buffers (images + descriptor table), alternated per batch; an event recorded behind each
        # upload guards the buffer's reuse, so nothing on the batch path waits for the training kernels queued on the stream
        self._pinned = [None, None]
        self._copied = [None, None]
        self._turn = 0

    def _kmax(self, crop):
        support = 2.0 if self.filter == 1 else 1.0
        return int(math.ceil(support * max(1.0, crop / self.size))) * 2 + 1

    def __call__(self, images, params=None):
        B, S = len(images), self.size
        if params is None:
            params = [draw_train_params(S, self.color_jitter, self.reprob) if self.train else None for _ in images]
        descs = (hip.ImageDesc * B)()
        total = sum(int(im.shape[0]) * int(im.shape[1]) * 3 for im in images)
        total_al = (total + 255) // 256 * 256
        dsize = ctypes.sizeof(descs)
        slot = self._turn
        self._turn = 1 - slot
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()     # the upload that last read this buffer (two batches ago) has finished
        if self._pinned[slot] is None or self._pinned[slot].numel() < total_al + dsize:
            self._pinned[slot] = torch.empty(max(total_al + dsize, 1 << 20), dtype=torch.uint8).pin_memory()
        host = self._pinned[slot].numpy()
        off, max_crop, kmax = 0, 1, 3
        for i, (im, pr) in enumerate(zip(images, params)):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError("images must be uint8 HWC RGB arrays")
            h, w = int(im.shape[0]), int(im.shape[1])
            n = h * w * 3
            host[off:off + n] = im.reshape(-1)
            d = descs[i]
            d.src_offset, d.src_h, d.src_w = off, h, w
            if self.train:
                d.crop_top, d.crop_left, d.crop_h, d.crop_w = center_square_box(h, w)
            else:
                d.crop_top, d.crop_left, d.crop_h, d.crop_w = 0, 0, h, w   # Resize([s, s]): the whole image, squashed
            if pr is not None:
                d.hflip, d.vflip = int(pr["hflip"]), int(pr["vflip"])
                for k in range(3):
                    d.jitter_order[k] = int(pr["order"][k])
                    d.jitter_factor[k] = float(pr["factors"][k])
                d.erase_top, d.erase_left, d.erase_h, d.erase_w = (int(v) for v in pr["erase"])
                d.erase_seed = int(pr["seed"]) & 0xFFFFFFFF
            else:
                for k in range(3):
                    d.jitter_order[k] = -1
                    d.jitter_factor[k] = 1.0
            max_crop = max(max_crop, d.crop_h)
            kmax = max(kmax, self._kmax(d.crop_h), self._kmax(d.crop_w))
            off += n
        host[total_al:total_al + dsize] = np.frombuffer(bytes(descs), dtype=np.uint8)
        staged = torch.empty(total_al + dsize, dtype=torch.uint8, device=self.device)
        staged.copy_(self._pinned[slot][:total_al + dsize], non_blocking=True)      # one async upload: images + descriptors
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._copied[slot] = ev
        src, ddev = staged[:total], staged[total_al:]
        need = self.lib.icamd_image_pipeline_workspace_bytes(B, max_crop, S, S, kmax)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=self.device)
        hip.check(self.lib.icamd_image_pipeline(src.data_ptr(), ddev.data_ptr(), B, max_crop, S, S, self.filter, kmax,
                                                self.mean, self.std, out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                hip.stream_ptr()), "image_pipeline")
        self._last = (B, max_crop, kmax, src, ddev)     # keeps the inputs alive until the stream has consumed them
        return out

    def last_uint8(self):
        """uint8 [B, size, size, 3] image of the last call after resize / flips / jitter (parity tests)."""
        B, max_crop, kmax = self._last[:3]
        p = ctypes.c_void_p()
        hip.check(self.lib.icamd_image_pipeline_u8(self._ws.data_ptr(), B, max_crop, self.size, self.size, kmax, ctypes.byref(p)),
                  "image_pipeline_u8")
        base = self._ws.data_ptr()
        o = p.value - base
        return self._ws[o:o + B * self.size * self.size * 3].view(B, self.size, self.size, 3)


class GpuAugmentLoader:
    """Wraps a loader that yields (list of uint8 HWC arrays, targets) and hands the engine device batches."""

    def __init__(self, loader, pipeline):
        self.loader, self.pipeline = loader, pipeline

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, targets in self.loader:
            yield self.pipeline(images), torch.as_tensor(targets, dtype=torch.int64)

    def __getattr__(self, name):
        return getattr(self.__dict__["loader"], name)


def raw_collate(batch):
    """DataLoader collate for RawImageFolder: images stay a list of arrays (sizes differ), targets become a list."""
    return [b[0] for b in batch], [b[1] for b in batch]
