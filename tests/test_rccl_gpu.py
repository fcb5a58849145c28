"""The library's RCCL call site (include/icamd.h `icamd_rccl_*`, `icamd_allreduce_bucket_launch`) on the real
transport, with ONE rank: backend "nccl" (= RCCL on ROCm) is initialised at world size 1 and the gradient reducer's
bucket / side-stream / event chain is forced on, so every piece of the multi-GPU path except the wire itself runs on
the GPU box (reference: DistributedDataParallel wrap /root/reference/train.py:218-222, process group utils.py:339-375).
The multi-rank arithmetic (reduced gradient = mean of the ranks' gradients) is covered by tests/test_ddp_cpu.py and
tests/test_ddp_gpu.py over gloo."""
import ctypes
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module")
def nccl_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_c_abi_allreduce_and_broadcast(nccl_world1):
    from imageclassification_amd import ddp, hip
    lib = hip.load()
    assert lib.icamd_rccl_available() == 1 and lib.icamd_rccl_version() > 0
    comm = ddp.RcclComm()
    assert comm.info() == (1, 0)
    side = torch.cuda.Stream()
    x = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
    ref = x.clone()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    comm.all_reduce(x, ddp.RED_SUM, side.cuda_stream)          # sum over one rank: identity, through RCCL's kernel
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    comm.all_reduce(flag, ddp.RED_MIN, side.cuda_stream)
    comm.broadcast(x[:1024], 0, side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and int(flag.item()) == 0
    # bad arguments are reported, not launched
    assert lib.icamd_allreduce_bucket_launch(comm.handle, x.data_ptr(), 0, 0, 0, side.cuda_stream) != 0
    assert lib.icamd_allreduce_bucket_launch(comm.handle, x.data_ptr(), 16, 9, 0, side.cuda_stream) != 0
    assert lib.icamd_allreduce_bucket_launch(None, x.data_ptr(), 16, 0, 0, side.cuda_stream) != 0
    comm.destroy()


def test_reducer_on_rccl_matches_the_undistributed_step(nccl_world1):
    """Two training steps of ResNet-18 with the reducer forced on over RCCL (world 1: every bucket all-reduce, the flag
    MIN-reduce, the constructor broadcast and the side-stream events really run) are bit-identical to the same steps
    without it, and the buckets go out in reverse arena order while backward is still running."""
    from imageclassification_amd.ddp import DistributedDataParallel
    from imageclassification_amd.engine import train_one_epoch
    from imageclassification_amd.mixup import LabelSmoothingCrossEntropy
    from imageclassification_amd.nets import ResNet
    from imageclassification_amd.optim_factory import create_optimizer
    from imageclassification_amd.utils import NativeScalerWithGradNormCount
    C, B = 10, 8
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(B, 3, 64, 64, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(2)]

    def run(wrap):
        net = ResNet("resnet18", C, seed=11)
        model = DistributedDataParallel(net, force=True, first_bucket_mb=0.25, bucket_mb=8.0, last_bucket_mb=1.0) if wrap else net
        opt = create_optimizer("adamw", 1e-3, 5e-4, net)
        train_one_epoch(model, LabelSmoothingCrossEntropy(0.1), data, opt, DEV, 0, NativeScalerWithGradNormCount(), None,
                        None, None, start_steps=0, lr_schedule_values=[1e-3, 1e-3], wd_schedule_values=[5e-4, 5e-4],
                        num_training_steps_per_epoch=2, update_freq=1, use_amp=True, num_classes=C)
        torch.cuda.synchronize()
        return net, model

    plain, _ = run(False)
    net, model = run(True)
    red = model.reducer
    assert red.transport == "rccl" and red.comm is not None and red.ranks_seen() == 1
    assert len(red.buckets) >= 4
    assert torch.equal(net.param_arena, plain.param_arena)
    assert torch.equal(net.grad_arena, plain.grad_arena)
    # non-finite loss: the flag travels through the MIN all-reduce and the step is dropped
    before = net.param_arena.clone()
    bad = [(data[0][0].clone(), data[0][1])]
    bad[0][0][0, 0, 0, 0] = float("inf")
    opt = create_optimizer("adamw", 1e-3, 5e-4, net)
    train_one_epoch(model, LabelSmoothingCrossEntropy(0.1), bad, opt, DEV, 0, NativeScalerWithGradNormCount(), None, None,
                    None, start_steps=0, lr_schedule_values=[1e-2], wd_schedule_values=[0.0],
                    num_training_steps_per_epoch=1, update_freq=1, use_amp=True, num_classes=C)
    assert torch.equal(net.param_arena, before) and opt.steps_taken == 0
