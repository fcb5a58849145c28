"""Checkpoint save / resume with the reference's file and key layout (/root/reference/utils.py:536-615):
./train_cls/output/checkpoint-{epoch|best|best-ema}.pth holding keys model / optimizer / epoch / scaler /
input_shape / num_classes / args (+ model_ema).  The reference pickles the whole nn.Module under "model" and
consumers call `checkpoint["model"].state_dict()` (utils.py:582, val.py, modelchange.py); a model that owns
device arenas and a ctypes handle cannot be pickled, so "model" holds a small picklable snapshot object with
the same `.state_dict()` method (timm parameter names, torch layouts)."""
import glob
import os
import re

import torch

from . import utils

OUTPUT_DIR = os.path.join(".", "train_cls", "output")


class ModelSnapshot:
    """Picklable stand-in for the pickled module: `.state_dict()` returns CPU tensors in torch layout."""

    def __init__(self, arch, num_classes, state):
        self.arch, self.num_classes, self._state = arch, num_classes, state

    def state_dict(self):
        return self._state


def save_model(args, epoch, model, optimizer, loss_scaler, input_shape, num_classes, model_ema=None, output_dir=None):
    if not utils.is_main_process():
        return
    out = output_dir or OUTPUT_DIR
    os.makedirs(out, exist_ok=True)
    to_save = {"model": ModelSnapshot(model.arch, num_classes, model.state_dict()), "optimizer": optimizer.state_dict(),
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
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    src = ckpt["model"].state_dict()
    own = model_without_ddp.state_dict()
    kept = {k: v for k, v in src.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    mismatched = [k for k in own if k not in kept]
    own.update(kept)
    model_without_ddp.load_state_dict(own)
    print("Resume checkpoint %s (%d/%d tensors matched)" % (args.resume, len(kept), len(own)))
    if model_ema is not None:
        if "model_ema" in ckpt:
            esd = model_ema.module.state_dict()
            esd.update({k: v for k, v in ckpt["model_ema"].items() if k in esd and tuple(v.shape) == tuple(esd[k].shape)})
            model_ema.module.load_state_dict(esd)
        else:
            model_ema.set(model_without_ddp)
    if not mismatched and "optimizer" in ckpt and "epoch" in ckpt and isinstance(ckpt["epoch"], int):
        optimizer.load_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:
            loss_scaler.load_state_dict(ckpt["scaler"])
        print("With optim & sched!")
