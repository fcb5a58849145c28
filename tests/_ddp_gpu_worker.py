"""Worker of tests/test_ddp_gpu.py: one of two ranks sharing the single GPU (gloo carries the collectives, so the test
needs no second device; the product's reducer, side streams and hooks are exactly the ones the RCCL path uses).

Checks, per model family: the gradients the bucketed/overlapped reducer leaves in the arena, times 1/world, equal the mean
of the ranks' local gradients (computed without the reducer on the same batch); and after two train_one_epoch steps every
rank holds bit-identical parameters."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(arch, C):
    if arch.startswith("resnet"):
        from imageclassification_amd.nets import ResNet
        return ResNet(arch, C, seed=5), 64
    if arch.startswith("vit"):
        from imageclassification_amd.vit import VisionTransformer
        return VisionTransformer(arch, C, img_size=32, seed=5), 32
    from imageclassification_amd.convnext import ConvNeXt
    return ConvNeXt(arch, C, drop_path_rate=0.0, seed=5), 64


def main():
    """argv: [--transport torch|rccl] arch...   env ICAMD_RANK_BACKEND: gloo (default: both ranks share GPU 0) or nccl (one GPU
    per rank, RCCL: tests/test_rccl_gpu.py::test_two_gpu_*)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("ICAMD_RANK_BACKEND", "gloo")
    argv = sys.argv[1:]
    transport = None
    if argv and argv[0] == "--transport":
        transport, argv = argv[1], argv[2:]
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    from imageclassification_amd import hip
    from imageclassification_amd.ddp import DistributedDataParallel
    from imageclassification_amd.engine import train_one_epoch
    from imageclassification_amd.mixup import LabelSmoothingCrossEntropy
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    lib = hip.load()
    C, B = 10, 8
    for arch in argv:
        net, hw = build(arch, C)
        ddp = DistributedDataParallel(net, first_bucket_mb=0.05, bucket_mb=0.5, transport=transport)   # many buckets on a small model
        assert len(ddp.reducer.buckets) >= 3, len(ddp.reducer.buckets)
        if transport is not None:
            assert ddp.reducer.transport == transport and ddp.reducer.ranks_seen() == world
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(B, 3, hw, hw, generator=g).to(dev)
        y = torch.randint(0, C, (B,), generator=g).to(dev)

        def fwd_bwd():
            ws = net.pack(x)
            logits = net.forward_packed(ws)
            hip.check(lib.icamd_softmax_xent(logits.data_ptr(), net.ncls_p, B, C, y.data_ptr(), None, 1.0, 0.1, 1.0 / B,
                                             ws["loss_rows"].data_ptr(), ws["pred"].data_ptr(), ws["dlogits"].data_ptr(),
                                             hip.stream_ptr()), "xent")
            net.backward_packed(ws)

        hook = net.grad_ready_hook
        net.grad_ready_hook = None
        fwd_bwd()
        torch.cuda.synchronize()
        local = net.grad_arena.clone()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(t.double() for t in gathered) / world
        net.grad_ready_hook = hook
        ddp.reducer.reset()
        fwd_bwd()
        ddp.reducer.finish()
        torch.cuda.synchronize()
        got = net.grad_arena.double() * ddp.reducer.grad_scale
        err = (got - mean).abs().max().item()
        scale = mean.abs().max().item()
        assert err <= 1e-6 * max(scale, 1e-3), (arch, err, scale)
        assert sorted(ddp.reducer.launched) == list(range(len(ddp.reducer.buckets)))

        # two optimizer steps through the drop-in boundary: ranks stay bit-identical, and the per-bucket optimizer
        # on the reducer's side stream leaves bit for bit what the single launch behind the last bucket leaves -- parameters,
        # both Adam moments, the EMA arena and the BatchNorm buffers -- on the same two batches from the same start
        from imageclassification_amd.ema import ModelEmaV3
        opt = create_optimizer("adamw", 1e-3, 0.05, net)
        ema = ModelEmaV3(net, decay=0.9)
        data = [(torch.randn(B, 3, hw, hw, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(2)]
        start = (net.param_arena.clone(), net.buffer_arena.clone(), net.num_batches_tracked)
        results = {}
        for mode in ("1", "0"):
            os.environ["ICAMD_BUCKET_OPTIM"] = mode
            net.param_arena.copy_(start[0]); net.buffer_arena.copy_(start[1]); net.num_batches_tracked = start[2]
            net.refresh_shadow()
            opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt._load_step(0)
            ema.set(net)
            ddp.reducer.reset()
            train_one_epoch(ddp, LabelSmoothingCrossEntropy(0.1), data, opt, dev, 0, NativeScalerWithGradNormCount(), None, ema,
                            None, start_steps=0, lr_schedule_values=[1e-3, 1e-3], wd_schedule_values=[0.05, 0.05],
                            num_training_steps_per_epoch=2, update_freq=1, use_amp=True, num_classes=C)
            torch.cuda.synchronize()
            results[mode] = [t.clone() for t in (net.param_arena, opt.exp_avg, opt.exp_avg_sq, ema.param_arena, net.buffer_arena,
                                                 ema.module.buffer_arena)]
            if mode == "1":
                # the optimizer ran as one launch per bucket, each behind that bucket's all-reduce on the side stream
                assert ddp.reducer.callbacks == ddp.reducer.launched and len(ddp.reducer.callbacks) == 2 * len(ddp.reducer.buckets)
            else:
                assert not ddp.reducer.callbacks, ddp.reducer.callbacks
            assert opt.steps_taken == 2
        os.environ.pop("ICAMD_BUCKET_OPTIM")
        for name, a, b in zip(("params", "exp_avg", "exp_avg_sq", "ema params", "buffers", "ema buffers"), results["1"], results["0"]):
            assert torch.equal(a, b), (arch, "per-bucket optimizer differs from the single launch in", name)
        assert not torch.equal(results["1"][0], start[0])
        mine = net.param_arena.clone()
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1]), arch

        # a non-finite loss on ONE rank: the MIN-reduced flag makes BOTH ranks drop the step and count it as dropped
        before = net.param_arena.clone()
        bad = data[0][0].clone()
        if rank == 1:
            bad[0, 0, 0, 0] = float("inf")
        stats = train_one_epoch(ddp, LabelSmoothingCrossEntropy(0.1), [(bad, data[0][1])], opt, dev, 0,
                                NativeScalerWithGradNormCount(), None, None, None, start_steps=0, lr_schedule_values=[1e-2],
                                wd_schedule_values=[0.0], num_training_steps_per_epoch=1, update_freq=1, use_amp=True,
                                num_classes=C)
        torch.cuda.synchronize()
        assert torch.equal(net.param_arena, before), (arch, "step not dropped on rank", rank)
        assert opt.steps_taken == 2 and stats == {}, (opt.steps_taken, stats)    # no step counted in the meters on ANY rank
        ddp.shutdown()
        if rank == 0:
            print(f"ddp-ok {arch} buckets={len(ddp.reducer.buckets)} grad_err={err:.2e}", flush=True)
    dist.barrier()
    if rank == 0 and backend == "nccl":
        print(f"rccl2-ok {transport}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
