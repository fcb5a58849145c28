"""Command-line entry point with the reference's flags (/root/reference/train.py:32-107) and wiring order
(:110-407), driving the MI355X step engine.  `python train.py ...` on one GPU, or
`python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...` for one process per GPU.

Differences from the reference, all outside the hot path: models come from this package's registry
(resnet18/34/50) instead of timm.create_model, `--pretrained` defaults to False (no network), TensorBoard / W&B
sinks are not wired (the reference's need tensorboardX / wandb), and `--synthetic N` swaps the ImageFolder for
N seeded random images (BASELINE.json's timed configurations)."""
import argparse
import datetime
import json
import os
import time
from pathlib import Path

import numpy as np
import torch

from imageclassification_amd import utils
from imageclassification_amd.checkpoint import auto_load_model, save_model
from imageclassification_amd.datasets import SyntheticDataset, build_dataset
from imageclassification_amd.ddp import DistributedDataParallel
from imageclassification_amd.ema import ModelEmaV3
from imageclassification_amd.engine import evaluate, train_one_epoch
from imageclassification_amd.mixup import CrossEntropyLoss, LabelSmoothingCrossEntropy, Mixup, SoftTargetCrossEntropy
from imageclassification_amd.nets import ARCHS, ResNet
from imageclassification_amd.vit import CONFIGS as VIT_CONFIGS, VisionTransformer
from imageclassification_amd.convnext import CONFIGS as CNX_CONFIGS, ConvNeXt
from imageclassification_amd.optim_factory import create_optimizer
from imageclassification_amd.utils import NativeScalerWithGradNormCount as NativeScaler


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_args_parser():
    p = argparse.ArgumentParser("Training and evaluation script for image classification", add_help=False)
    a = p.add_argument
    a("--batch_size", default=64, type=int); a("--epochs", default=100, type=int); a("--update_freq", default=1, type=int)
    a("--pretrained", default=False, type=str2bool); a("--model", default="resnet50", type=str, metavar="MODEL")
    a("--drop_path", type=float, default=0.05, metavar="PCT"); a("--input_size", default=224, type=int)
    a("--model_ema", type=str2bool, default=False)
    a("--opt", default="adamw", type=str); a("--opt_eps", default=1e-8, type=float); a("--opt_betas", default=None, type=float)
    a("--clip_grad", type=float, default=None); a("--weight_decay", type=float, default=5e-4)
    a("--weight_decay_end", type=float, default=5e-6); a("--lr", type=float, default=1e-3)
    a("--min_lr", type=float, default=1e-6); a("--warmup_epochs", type=int, default=5); a("--warmup_steps", type=int, default=-1)
    a("--RASampler", default=False, type=str2bool); a("--color_jitter", type=float, default=0.3); a("--aa", type=str, default="")
    a("--smoothing", type=float, default=0.1)
    a("--reprob", type=float, default=0.25, metavar="PCT"); a("--remode", type=str, default="pixel")
    a("--recount", type=int, default=1); a("--resplit", type=str2bool, default=False)
    a("--mixup", type=float, default=0.8); a("--cutmix", type=float, default=0.0)
    a("--cutmix_minmax", type=float, nargs="+", default=None); a("--mixup_prob", type=float, default=1.0)
    a("--mixup_switch_prob", type=float, default=0.5); a("--mixup_mode", type=str, default="batch")
    a("--data_path", default="../../datas/CatsDogs_mini", type=str); a("--train_split_rato", default=0.9, type=float)
    a("--device", default="cuda"); a("--seed", default=88, type=int); a("--resume", default="")
    a("--auto_resume", type=str2bool, default=True); a("--save_ckpt", type=str2bool, default=True)
    a("--save_ckpt_freq", default=1, type=int); a("--save_ckpt_num", default=999, type=int)
    a("--start_epoch", default=0, type=int); a("--eval", type=str2bool, default=False)
    a("--num_workers", default=32, type=int); a("--use_amp", type=str2bool, default=False)
    a("--world_size", default=1, type=int); a("--local_rank", default=-1, type=int)
    a("--dist_on_itp", type=str2bool, default=False); a("--dist_url", default="env://")
    a("--enable_wandb", type=str2bool, default=False); a("--project", default="classification", type=str)
    a("--wandb_ckpt", type=str2bool, default=False)
    # additions for the MI355X build
    a("--synthetic", default=0, type=int, help="use N seeded random images instead of --data_path")
    a("--gpu_aug", type=str2bool, default=False,
      help="run the transforms of datasets.py (resize, flips, colour jitter, normalise, random erasing) as HIP kernels on the "
           "decoded uint8 images instead of per sample on the host (imageclassification_amd/gpu_pipeline.py)")
    a("--num_classes", default=1000, type=int, help="classes of the synthetic dataset")
    return p


def create_model(name, num_classes, input_size=224, drop_path=0.0):
    if name in ARCHS:
        return ResNet(name, num_classes)
    if name in VIT_CONFIGS:
        return VisionTransformer(name, num_classes, img_size=input_size)
    if name in CNX_CONFIGS:
        return ConvNeXt(name, num_classes, drop_path_rate=drop_path)   # reference train.py:189-192
    raise ValueError(f"model '{name}' is not built for the MI355X path yet (available: {sorted(ARCHS) + sorted(VIT_CONFIGS) + sorted(CNX_CONFIGS)})")


def main(args):
    utils.init_distributed_mode(args)
    print(args)
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)

    if args.synthetic:
        num_classes = args.num_classes
        dataset_train = SyntheticDataset(args.synthetic, num_classes, args.input_size, seed=args.seed)
        dataset_val = SyntheticDataset(max(args.synthetic // 8, args.batch_size), num_classes, args.input_size, seed=args.seed + 1)
    else:
        dataset_train, dataset_val, num_classes = build_dataset(args=args)

    num_tasks, global_rank = utils.get_world_size(), utils.get_rank()
    if args.RASampler:   # reference train.py:125-128
        sampler_train = utils.RASampler(dataset_train, num_replicas=num_tasks, rank=global_rank, shuffle=True)
    else:
        sampler_train = torch.utils.data.DistributedSampler(dataset_train, num_replicas=num_tasks, rank=global_rank,
                                                            shuffle=True, seed=args.seed)
    sampler_val = torch.utils.data.SequentialSampler(dataset_val)
    gpu_aug = bool(getattr(args, "gpu_aug", False)) and not args.synthetic
    if gpu_aug:   # workers only decode; the transforms run on the device, one C-ABI call per batch
        from imageclassification_amd.gpu_pipeline import GpuAugmentLoader, GpuImagePipeline, raw_collate
        data_loader_train = GpuAugmentLoader(
            torch.utils.data.DataLoader(dataset_train, sampler=sampler_train, batch_size=args.batch_size,
                                        num_workers=args.num_workers, drop_last=True, collate_fn=raw_collate),
            GpuImagePipeline(args.input_size, True, args.color_jitter, args.reprob, device=str(device)))
        data_loader_val = GpuAugmentLoader(
            torch.utils.data.DataLoader(dataset_val, sampler=sampler_val, batch_size=int(1.5 * args.batch_size),
                                        num_workers=args.num_workers, collate_fn=raw_collate),
            GpuImagePipeline(args.input_size, False, device=str(device)))
    else:
        data_loader_train = torch.utils.data.DataLoader(dataset_train, sampler=sampler_train, batch_size=args.batch_size,
                                                        num_workers=args.num_workers, pin_memory=True, drop_last=True)
        data_loader_val = torch.utils.data.DataLoader(dataset_val, sampler=sampler_val, batch_size=int(1.5 * args.batch_size),
                                                      num_workers=args.num_workers, pin_memory=True)
    input_shape = [1, 3, args.input_size, args.input_size]

    mixup_fn = None
    if args.mixup > 0 or args.cutmix > 0.0 or args.cutmix_minmax is not None:
        print("Mixup is activated!")
        mixup_fn = Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, cutmix_minmax=args.cutmix_minmax,
                         prob=args.mixup_prob, switch_prob=args.mixup_switch_prob, mode=args.mixup_mode,
                         label_smoothing=args.smoothing, num_classes=num_classes)

    model = create_model(args.model, num_classes, args.input_size, args.drop_path)
    model_ema = ModelEmaV3(model, decay=0.9995, device=device) if args.model_ema else None
    model_without_ddp = model
    n_parameters = sum(int(np.prod(p.torch_shape)) for p in model.params.values())
    print("number of params:", n_parameters)
    total_batch_size = args.batch_size * args.update_freq * utils.get_world_size()
    num_training_steps_per_epoch = len(dataset_train) // total_batch_size
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Update frequent = %d" % args.update_freq)
    print("Number of training examples = %d" % len(dataset_train))
    print("Number of training training per epoch = %d" % num_training_steps_per_epoch)
    if args.distributed:
        model = DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=False)
        model_without_ddp = model.module
    optimizer = create_optimizer(opt=args.opt, lr=args.lr, weight_decay=args.weight_decay, model=model_without_ddp)
    loss_scaler = NativeScaler()
    print("Use Cosine LR scheduler")
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs,
                                                num_training_steps_per_epoch)
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd_schedule_values), min(wd_schedule_values)))
    if mixup_fn is not None:
        criterion = SoftTargetCrossEntropy()
    elif args.smoothing > 0.0:
        criterion = LabelSmoothingCrossEntropy(smoothing=args.smoothing)
    else:
        criterion = CrossEntropyLoss()
    print("criterion = %s" % str(criterion))
    auto_load_model(args=args, model_without_ddp=model_without_ddp, optimizer=optimizer, loss_scaler=loss_scaler,
                    model_ema=model_ema)

    if args.eval:
        print("Eval only mode")
        target = model_ema.module if args.model_ema else model
        test_stats = evaluate(data_loader_val, target, device, num_classes=num_classes, use_amp=args.use_amp)
        print(f"Accuracy of the network on {len(dataset_val)} test images: {test_stats['acc1']:.5f}%")
        return test_stats

    max_accuracy, max_accuracy_ema = 0.0, 0.0
    print("Start training for %d epochs" % args.epochs)
    start_time = time.time()
    log_stats = {}
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            data_loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, criterion, data_loader_train, optimizer, device, epoch, loss_scaler,
                                      args.clip_grad, model_ema, mixup_fn, log_writer=None, wandb_logger=None,
                                      start_steps=epoch * num_training_steps_per_epoch,
                                      lr_schedule_values=lr_schedule_values, wd_schedule_values=wd_schedule_values,
                                      num_training_steps_per_epoch=num_training_steps_per_epoch,
                                      update_freq=args.update_freq, use_amp=args.use_amp, num_classes=num_classes)
        ckpt = dict(args=args, input_shape=input_shape, model=model_without_ddp, optimizer=optimizer,
                    loss_scaler=loss_scaler, model_ema=model_ema, num_classes=num_classes)
        if args.save_ckpt and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            save_model(epoch=epoch, **ckpt)
        test_stats = evaluate(data_loader_val, model, device, num_classes=num_classes, use_amp=args.use_amp)
        print(f"Accuracy of the model on the {len(dataset_val)} test images: {test_stats['acc1']:.3f}%")
        if max_accuracy < test_stats["acc1"]:
            max_accuracy = test_stats["acc1"]
            if args.save_ckpt:
                save_model(epoch="best", **ckpt)
        print(f"Max accuracy: {max_accuracy:.3f}%")
        log_stats = {"current_time": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"),
                     **{f"train_{k}": v for k, v in train_stats.items()},
                     **{f"test_{k}": v for k, v in test_stats.items()},
                     "epoch": epoch, "n_parameters": f"{n_parameters / 1e6:.2f}M"}
        if args.model_ema:
            test_stats_ema = evaluate(data_loader_val, model_ema.module, device, num_classes, use_amp=args.use_amp)
            print(f"Accuracy of the model EMA on {len(dataset_val)} test images: {test_stats_ema['acc1']:.1f}%")
            if max_accuracy_ema < test_stats_ema["acc1"]:
                max_accuracy_ema = test_stats_ema["acc1"]
                if args.save_ckpt:
                    save_model(epoch="best-ema", **ckpt)
                print(f"Max EMA accuracy: {max_accuracy_ema:.2f}%")
            log_stats.update({f"test_{k}_ema": v for k, v in test_stats_ema.items()})
        if utils.is_main_process():
            with open(os.path.join("train_cls", "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    total_time_str = str(datetime.timedelta(seconds=int(time.time() - start_time)))
    print("Training time {}".format(total_time_str))
    return log_stats


if __name__ == "__main__":
    parser = argparse.ArgumentParser("Classification training and evaluation script", parents=[get_args_parser()])
    args = parser.parse_args()
    Path("./train_cls/output").mkdir(parents=True, exist_ok=True)
    main(args)
