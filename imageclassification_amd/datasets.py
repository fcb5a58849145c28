"""ImageFolder input pipeline (reference surface: /root/reference/datasets.py:57-144, loaders train.py:152-170).

Boundary only: decoding and augmentation stay on the host (PIL + numpy; torchvision / timm are not available
here), the device path starts at the fp32 NCHW batch handed to train_one_epoch.  Eval transform =
Resize([s, s]) -> ToTensor -> Normalize(ImageNet mean/std) (datasets.py:139-144).  Train transform = the
subset of timm.create_transform the reference configures (datasets.py:124-136: RandomResizedCrop with scale=(1,1),
ratio=(1,1) -- the whole image when it is square, timm's fallback centre crop to min(W, H) otherwise -- resized
bicubically to the input size, hflip 0.5, vflip 0.5, colour jitter, pixel-mode random erasing); auto-augment policies
(`--aa`) are not implemented and raise.
"""
import json
import os
import random

import numpy as np
import torch
from PIL import Image

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolder(torch.utils.data.Dataset):
    """root/<class>/<image>: classes sorted by name -> index, samples sorted by path (torchvision's convention)."""

    def __init__(self, root, transform=None):
        self.root = root
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"no class folders under {root}")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        self.targets = [t for _, t in self.samples]
        self.transform = transform

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        path, target = self.samples[i]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")
        return (self.transform(img) if self.transform else img), target


class RawSubset(torch.utils.data.Dataset):
    """Decoded but untransformed samples (uint8 HWC arrays) for the GPU input pipeline (gpu_pipeline.py)."""

    def __init__(self, base, indices=None):
        self.base = base
        self.indices = list(range(len(base.samples))) if indices is None else list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        path, target = self.base.samples[self.indices[i]]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")
        return np.asarray(img, dtype=np.uint8), target


class _Subset(torch.utils.data.Dataset):
    def __init__(self, base, indices, transform):
        self.base, self.indices, self.transform = base, list(indices), transform

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        path, target = self.base.samples[self.indices[i]]
        with open(path, "rb") as f:
            img = Image.open(f).convert("RGB")
        return self.transform(img), target


def _to_tensor_normalized(img, mean, std):
    a = np.asarray(img, dtype=np.float32) / 255.0
    a = (a - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


class EvalTransform:
    def __init__(self, input_size):
        self.size = input_size

    def __call__(self, img):
        img = img.resize((self.size, self.size), Image.BILINEAR)   # torchvision Resize default interpolation
        return _to_tensor_normalized(img, IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)


class TrainTransform:
    def __init__(self, input_size, color_jitter=0.3, reprob=0.25, vflip=0.5, hflip=0.5, auto_augment=""):
        if auto_augment:
            raise NotImplementedError("--aa auto-augment policies need timm and are outside the MI355X hot path")
        self.size, self.cj, self.reprob, self.vflip, self.hflip = input_size, color_jitter, reprob, vflip, hflip

    def _jitter(self, a):
        ops = [0, 1, 2]
        random.shuffle(ops)
        for op in ops:
            f = random.uniform(max(0.0, 1 - self.cj), 1 + self.cj)
            gray = (a * np.asarray([0.299, 0.587, 0.114], dtype=np.float32)).sum(-1, keepdims=True)
            if op == 0:
                a = a * f
            elif op == 1:
                a = (a - gray.mean()) * f + gray.mean()
            else:
                a = (a - gray) * f + gray
            a = np.clip(a, 0.0, 1.0)
        return a

    def __call__(self, img):
        W, H = img.size
        if W != H:      # scale=(1,1), ratio=(1,1) can only be met by the fallback: a centred square of side min(W, H)
            side = min(W, H)
            left, top = (W - side) // 2, (H - side) // 2
            img = img.crop((left, top, left + side, top + side))
        img = img.resize((self.size, self.size), Image.BICUBIC)
        a = np.asarray(img, dtype=np.float32) / 255.0
        if random.random() < self.hflip:
            a = a[:, ::-1]
        if random.random() < self.vflip:
            a = a[::-1]
        if self.cj and self.cj > 0:
            a = self._jitter(a)
        a = (a - np.asarray(IMAGENET_DEFAULT_MEAN, dtype=np.float32)) / np.asarray(IMAGENET_DEFAULT_STD, dtype=np.float32)
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        if self.reprob > 0 and random.random() < self.reprob:   # pixel-mode random erasing, one box
            area = self.size * self.size
            for _ in range(10):
                target = random.uniform(0.02, 1 / 3) * area
                aspect = np.exp(random.uniform(np.log(0.3), np.log(1 / 0.3)))
                h, w = int(round(np.sqrt(target * aspect))), int(round(np.sqrt(target / aspect)))
                if 0 < h < self.size and 0 < w < self.size:
                    top, left = random.randint(0, self.size - h), random.randint(0, self.size - w)
                    t[:, top:top + h, left:left + w] = torch.randn(3, h, w)
                    break
        return t


def build_transform(is_train, args):
    if is_train:
        return TrainTransform(args.input_size, args.color_jitter, args.reprob, auto_augment=getattr(args, "aa", ""))
    return EvalTransform(args.input_size)


def split_dataset(dataset, train_ratio):
    """Per-class split with the SAME number of validation images for every class (reference datasets.py:12-53);
    uses Python's `random`, which the reference leaves unseeded (SURVEY Appx C.11)."""
    by_class = {}
    for i, t in enumerate(dataset.targets):
        by_class.setdefault(t, []).append(i)
    n_min = min(len(v) for v in by_class.values())
    n_val = n_min - int(n_min * train_ratio)             # reference datasets.py:25
    train_idx, val_idx = [], []
    for t, idxs in sorted(by_class.items()):
        idxs = list(idxs)
        random.shuffle(idxs)
        # reference datasets.py:29-30: the LAST n_val of the shuffled list validate (n_val == 0 reproduces its
        # `indices[:-0]` / `indices[-0:]` quirk: everything validates, nothing trains)
        train_idx += idxs[:-n_val] if n_val else []
        val_idx += idxs[-n_val:] if n_val else idxs
    return train_idx, val_idx


def _write_class_indices(class_to_idx, output_dir=None):
    """./train_cls/output/class_indices.json = {index: class name} (reference datasets.py:93-97,110-114; read by the
    reference's inference tools)."""
    out = output_dir or os.path.join(".", "train_cls", "output")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "class_indices.json"), "w") as f:
        f.write(json.dumps(dict((val, key) for key, val in class_to_idx.items()), indent=4))


def build_dataset(args):
    """Returns (train_dataset, val_dataset, num_classes). train_split_rato == 0 -> data_path/train and /val.
    args.gpu_aug: the datasets yield decoded uint8 arrays and the transforms run on the GPU (gpu_pipeline.py)."""
    ratio = getattr(args, "train_split_rato", 0.9)
    if getattr(args, "gpu_aug", False):
        if ratio == 0:
            tr_base, va_base = ImageFolder(os.path.join(args.data_path, "train")), ImageFolder(os.path.join(args.data_path, "val"))
            _write_class_indices(tr_base.class_to_idx)
            print("Number of the class = %d" % len(tr_base.classes))
            return RawSubset(tr_base), RawSubset(va_base), len(tr_base.classes)
        base = ImageFolder(args.data_path)
        tr, va = split_dataset(base, ratio)
        _write_class_indices(base.class_to_idx)
        print("Number of the class = %d" % len(base.classes))
        return RawSubset(base, tr), RawSubset(base, va), len(base.classes)
    t_train, t_val = build_transform(True, args), build_transform(False, args)
    if ratio == 0:
        train = ImageFolder(os.path.join(args.data_path, "train"), t_train)
        val = ImageFolder(os.path.join(args.data_path, "val"), t_val)
        _write_class_indices(train.class_to_idx)
        print("Number of the class = %d" % len(train.classes))
        return train, val, len(train.classes)
    base = ImageFolder(args.data_path)
    tr, va = split_dataset(base, ratio)
    _write_class_indices(base.class_to_idx)
    print("Number of the class = %d" % len(base.classes))
    return _Subset(base, tr, t_train), _Subset(base, va, t_val), len(base.classes)


class SyntheticDataset(torch.utils.data.Dataset):
    """N(0,1) images and uniform labels from a fixed seed (BASELINE.json's timed configs are synthetic)."""

    def __init__(self, n, num_classes, input_size, seed=88):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(min(n, 512), 3, input_size, input_size, generator=g)
        self.y = torch.randint(0, num_classes, (n,), generator=g)
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.x[i % self.x.shape[0]], int(self.y[i])
