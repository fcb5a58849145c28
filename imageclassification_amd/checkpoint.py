"""Checkpoint save / resume with the reference's file and key layout (/root/reference/utils.py:536-615):
./train_cls/output/checkpoint-{epoch|best|best-ema}.pth holding keys model / optimizer / epoch / scaler /
input_shape / num_classes / args (+ model_ema).

The reference pickles the whole nn.Module under "model" (utils.py:542) and its consumers use the unpickled object as
a model: `checkpoint["model"](img)` after `.eval()` / `.to(device)` (val.py:14-28, also deep-copied by timm's
ModelEmaV3 there), `checkpoint["model"].load_state_dict(checkpoint["model_ema"])` followed by a re-save
(modelchange.py:155-162), `checkpoint["model"].state_dict()` (utils.py:582).  A HIP model owns device arenas and a
ctypes handle, so what is pickled is its recipe -- class, constructor arguments, CPU state_dict, mode -- and what
comes back from torch.load is a `DeferredModel`: it answers state_dict / load_state_dict / eval / train / to on the CPU
copy (no GPU needed: modelchange.py's EMA -> model conversion runs with map_location="cpu"), and builds the device
arenas on first use as a model (`__call__`, or any other attribute of the live class)."""
import glob
import importlib
import os
import re
from collections import OrderedDict

import torch

from . import utils

OUTPUT_DIR = os.path.join(".", "train_cls", "output")


def _rebuild_model(module, cls_name, kwargs, state, training):
    return DeferredModel(module, cls_name, kwargs, state, training)


class PicklableModel:
    """Mixin of the HIP model classes: `torch.save` / `pickle` / `copy.deepcopy` of a live model store its recipe."""

    def _ctor_kwargs(self):
        raise NotImplementedError

    def __reduce__(self):
        cls = type(self)
        state = OrderedDict((k, v.detach().cpu().clone()) for k, v in self.state_dict().items())
        return _rebuild_model, (cls.__module__, cls.__name__, self._ctor_kwargs(), state, bool(self.training))


class DeferredModel:
    """What `checkpoint["model"]` is after torch.load: the pickled recipe of a HIP model, materialised on first use."""

    def __init__(self, module, cls_name, kwargs, state, training):
        self.__dict__.update(_module=module, _cls_name=cls_name, _kwargs=dict(kwargs), _state=OrderedDict(state),
                             training=bool(training), _live=None, _device="cuda")

    # ---- the nn.Module surface the reference's consumers touch, served without a GPU
    def state_dict(self):
        return self._live.state_dict() if self._live is not None else OrderedDict(self._state)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._state if k not in sd]
        unexpected = [k for k in sd if k not in self._state]
        if strict and (missing or unexpected):
            raise KeyError(f"load_state_dict: missing {missing[:5]}, unexpected {unexpected[:5]}")
        for k, v in sd.items():
            if k in self._state:
                if tuple(v.shape) != tuple(self._state[k].shape):
                    raise ValueError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._state[k].shape)}")
                self._state[k] = v.detach().to("cpu", self._state[k].dtype).clone()
        if self._live is not None:
            self._live.load_state_dict(self._state)
        return missing

    def eval(self):
        return self.train(False)

    def train(self, mode=True):
        self.__dict__["training"] = bool(mode)
        if self._live is not None:
            self._live.train(mode)
        return self

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            self.__dict__["_device"] = str(device)
        return self

    def parameters(self):
        return self.materialise().parameters()

    # ---- first use as a model
    def materialise(self):
        if self._live is None:
            cls = getattr(importlib.import_module(self._module), self._cls_name)
            kw = dict(self._kwargs)
            dev = self._device if str(self._device).startswith("cuda") else "cuda"   # the compute path is GPU-only
            live = cls(device=dev, **kw)
            live.load_state_dict(self._state)
            live.train(self.training)
            self.__dict__["_live"] = live
        return self._live

    def __call__(self, x):
        return self.materialise()(x)

    def __getattr__(self, name):          # anything else of the live class (forward_packed, arch, num_classes, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        if name in self._kwargs:
            return self._kwargs[name]
        return getattr(self.materialise(), name)

    def __reduce__(self):
        return _rebuild_model, (self._module, self._cls_name, self._kwargs, self.state_dict(), self.training)


def save_model(args, epoch, model, optimizer, loss_scaler, input_shape, num_classes, model_ema=None, output_dir=None):
    if not utils.is_main_process():
        return
    out = output_dir or OUTPUT_DIR
    os.makedirs(out, exist_ok=True)
    # "model": the model object itself, as the reference does (utils.py:542); PicklableModel.__reduce__ stores its recipe
    to_save = {"model": model, "optimizer": optimizer.state_dict(),
               "epoch": epoch, "scaler": loss_scaler.state_dict(), "input_shape": input_shape,
               "num_classes": num_classes, "args": args}
    if model_ema is not None:
        to_save["model_ema"] = model_ema.state_dict()
    torch.save(to_save, os.path.join(out, f"checkpoint-{epoch}.pth"))
    keep = getattr(args, "save_ckpt_num", 999)
    if isinstance(epoch, int):
        old = epoch - getattr(args, "save_ckpt_freq", 1) * keep
        path = os.path.join(out, f"checkpoint-{old}.pth")
        if os.path.exists(path):
            os.remove(path)


def auto_load_model(args, model_without_ddp, optimizer, loss_scaler, model_ema=None, output_dir=None):
    out = output_dir or OUTPUT_DIR
    if getattr(args, "auto_resume", True) and not getattr(args, "resume", ""):
        latest = -1
        for f in glob.glob(os.path.join(out, "checkpoint-*.pth")):
            m = re.search(r"checkpoint-(\d+)\.pth$", f)
            if m:
                latest = max(latest, int(m.group(1)))
        if latest >= 0:
            args.resume = os.path.join(out, f"checkpoint-{latest}.pth")
        if getattr(args, "resume", ""):
            print("Auto resume checkpoint: %s" % args.resume)
    if not getattr(args, "resume", ""):
        return
    if str(args.resume).startswith("https"):
        # reference utils.py:576-578 downloads with torch.hub; this build runs without network access
        raise ValueError(f"--resume {args.resume}: URL checkpoints are not supported (no network); download the file and "
                         "pass its path")
    print(args.resume)
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    src = ckpt["model"].state_dict()
    own = model_without_ddp.state_dict()
    # reference utils.py:586-594: keep the SOURCE entries whose key and shape match, count the ones that do not
    kept, mismatched = {}, 0
    for k, v in src.items():
        if k in own and tuple(v.shape) == tuple(own[k].shape):
            kept[k] = v
        else:
            print(f"Skipping mismatched key: {k}")
            mismatched += 1
    own.update(kept)
    model_without_ddp.load_state_dict(own)      # strict=False semantics: entries the source lacks keep their values
    print("Resume checkpoint %s" % args.resume)
    if model_ema is not None:
        if "model_ema" in ckpt and mismatched == 0:
            model_ema.module.load_state_dict(ckpt["model_ema"])
        else:
            model_ema.set(model_without_ddp)
    if mismatched == 0 and "optimizer" in ckpt and "epoch" in ckpt and isinstance(ckpt["epoch"], int):
        optimizer.load_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:
            loss_scaler.load_state_dict(ckpt["scaler"])
        print("With optim & sched!")
