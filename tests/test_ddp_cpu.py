"""Data-parallel gradient reduction logic on CPU: bucket layout, launch order, averaging; world_size 2 over gloo."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bucket_layout_covers_arena_from_the_end():
    from imageclassification_amd.ddp import make_buckets
    n = 25_557_032 + 5000
    b = make_buckets(n, (1 << 20) // 4, (25 << 20) // 4)
    assert b[0][1] == n and b[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(b, b[1:]):
        assert hi2 == lo and lo % 64 == 0
    assert (b[0][1] - b[0][0]) * 4 <= (1 << 20) + 256            # small first bucket: communication starts early
    assert all((hi - lo) * 4 <= (25 << 20) + 256 for lo, hi in b)
    assert len(b) == 5                                          # ResNet-50: 97.5 MiB -> 1 MiB + 4 x 25 MiB
    # the bucket that holds the front of the arena (launched when backward ends: nothing overlaps it) is cut off small
    c = make_buckets(n, (1 << 20) // 4, (25 << 20) // 4, last_bucket_elems=(4 << 20) // 4)
    assert len(c) == 6 and c[-1] == (0, (4 << 20) // 4) and c[-2][0] == c[-1][1]
    assert c[:4] == b[:4] and sum(hi - lo for lo, hi in c) == n
    tiny = make_buckets(1000, 100, 300, align=1, last_bucket_elems=5000)   # cap larger than what is left: no extra cut
    assert tiny == make_buckets(1000, 100, 300, align=1)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imageclassification_amd.ddp import GradReducer
    from imageclassification_amd.utils import MetricLogger
    n = 300_000
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = GradReducer(flat, first_bucket_mb=0.1, bucket_mb=0.4)
    # backward reports completion from the end of the arena towards the start, in three chunks
    for lo in (250_000, 120_000, 0):
        red.grads_ready_from(lo)
    launched_before_finish = list(red.launched)
    # rank 1's loss was non-finite: after finish() EVERY rank sees the flag down and drops the step together
    flag = torch.tensor([1 if rank == 0 else 0], dtype=torch.int32)
    red.finish(flag)
    flag_ok = int(flag.item()) == 0 and red.transport == "torch" and red.ranks_seen() == 2
    other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    ok_sum = torch.allclose(flat, mine + other, rtol=1e-6, atol=1e-6)
    avg = flat * red.grad_scale
    ok_avg = torch.allclose(avg, (mine + other) / 2, rtol=1e-6, atol=1e-6)
    # meters: one packed float64 all-reduce
    ml = MetricLogger()
    ml.update(loss=1.0 + rank)
    ml.meters["acc1"].update(50.0 * (rank + 1), n=10 * (rank + 1))
    ml.synchronize_between_processes()
    meters_ok = (ml.meters["loss"].count == 2 and abs(ml.meters["loss"].global_avg - 1.5) < 1e-12 and
                 ml.meters["acc1"].count == 30 and abs(ml.meters["acc1"].total - (500.0 + 2000.0)) < 1e-9)
    q.put((rank, ok_sum, ok_avg, launched_before_finish, len(red.buckets), meters_ok and flag_ok))
    dist.destroy_process_group()


def test_two_rank_gloo_bucketed_allreduce_and_meter_sync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_sum, ok_avg, launched, nb, meters_ok in res:
        assert ok_sum and ok_avg and meters_ok
        assert launched == sorted(launched) and launched[0] == 0     # buckets go out in arena-end-first order
        assert 0 < len(launched) <= nb
