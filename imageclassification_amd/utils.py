"""Host-side collaborators of the step engine: meters, schedules, the loss-scaler protocol and the
distributed bootstrap.  Mirrors the reference's utils.py surface for the hot path only
(/root/reference/utils.py:65-204 meters, :311-375 distributed glue, :427-468 scaler, :471-488 schedule).
"""
import datetime
import math
import os
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist


class SmoothedValue:
    """A scalar series tracked two ways: a sliding window (median / mean / max / last) and a running
    sample-weighted total.  Same observable behaviour as the reference meter (utils.py:65-118): `update(v, n)`
    appends v once to the window and adds v*n to the total; `median` is torch.median's (lower middle value)."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt if fmt is not None else "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.count = 0
        self.total = 0.0

    def update(self, value, n=1):
        self.deque.append(value)
        self.total += value * n
        self.count += n

    # -- window statistics
    @property
    def median(self):
        window = sorted(float(v) for v in self.deque)
        return window[(len(window) - 1) // 2]

    @property
    def avg(self):
        return float(np.mean(np.asarray(self.deque, dtype=np.float32)))

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    # -- whole-series statistic
    @property
    def global_avg(self):
        return self.total / self.count

    def synchronize_between_processes(self):
        """Cross-rank sum of (count, total) in float64; the window stays rank-local (reference utils.py:80-88)."""
        if is_dist_avail_and_initialized():
            cnt, tot = all_reduce_f64([float(self.count), float(self.total)])
            self.count, self.total = int(cnt), tot

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    """Named SmoothedValue meters (reference utils.py:121-204)."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **scalars):
        for name, val in scalars.items():
            if val is None:
                continue
            if torch.is_tensor(val):
                val = val.item()
            if not isinstance(val, (int, float)):
                raise TypeError(f"meter '{name}' got a {type(val).__name__}, expected a number")
            self.meters[name].update(val)

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def __getattr__(self, name):
        meters = self.__dict__.get("meters", {})
        if name in meters:
            return meters[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        """ONE packed float64 all-reduce for all meters (the reference does a barrier + all-reduce per meter)."""
        if not is_dist_avail_and_initialized():
            return
        meters = list(self.meters.values())
        packed = []
        for m in meters:
            packed += [float(m.count), float(m.total)]
        packed = all_reduce_f64(packed)
        for i, m in enumerate(meters):
            m.count, m.total = int(packed[2 * i]), packed[2 * i + 1]

    def log_every(self, iterable, print_freq, header=None):
        """Yield from `iterable`, timing iterations; prints every `print_freq` items (0: only the final
        'Total time' line, which is how evaluate() uses it, reference engine.py:168)."""
        header = header or ""
        n = len(iterable) if hasattr(iterable, "__len__") else 0
        t_start = t_prev = time.time()
        it_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        for i, item in enumerate(iterable):
            data_time.update(time.time() - t_prev)
            yield item
            it_time.update(time.time() - t_prev)
            if print_freq and (i % print_freq == 0 or i == n - 1):
                eta = datetime.timedelta(seconds=int(it_time.global_avg * (n - i)))
                print(self.delimiter.join([header, f"[{i:{len(str(n))}d}/{n}]", f"eta: {eta}", str(self),
                                           f"time: {it_time}", f"data: {data_time}"]))
            t_prev = time.time()
        total = time.time() - t_start
        print("{} Total time: {} ({:.4f} s / it)".format(header, datetime.timedelta(seconds=int(total)),
                                                         total / max(n, 1)))


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0,
                     warmup_steps=-1):
    """Per-iteration schedule array (reference utils.py:471-488): linear warm-up from `start_warmup_value`
    (so the first step's value is 0 by default) followed by a half cosine to `final_value`."""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    print("Set warmup steps = %d" % warmup_iters)
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / (len(iters))))
                         for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


class NativeScalerWithGradNormCount:
    """The loss_scaler object train.py hands to train_one_epoch (reference utils.py:427-453).

    The reference scales an fp16 loss; the MI355X path computes in bf16 (fp32 range) and needs no scaling, so
    this object only carries the scaler state_dict key and the clip / grad-norm policy:
    `clip_grad is not None` -> clip by global norm, else just measure it (reference :438-442)."""
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._state = {"scale": 1.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                       "_growth_tracker": 0}

    def state_dict(self):
        return dict(self._state)

    def load_state_dict(self, state_dict):
        self._state.update(state_dict)


# ---------------------------------------------------------------------------------------------------
# distributed glue (reference utils.py:311-375): one process per GPU, RCCL over xGMI via torch.distributed
# ---------------------------------------------------------------------------------------------------
class RASampler(torch.utils.data.Sampler):
    """Repeated-augmentation sampler with the reference's exact index stream (/root/reference/utils.py:17-63; pinned by
    tests/golden/ra_sampler.json): the epoch-seeded permutation with every index tripled, wrapped to a multiple of the
    world size, dealt round-robin to the ranks, and cut to floor(len//256*256 / world) draws per rank."""

    REPEATS = 3

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None:
            num_replicas = get_world_size()
        if rank is None:
            rank = get_rank()
        self.n = len(dataset)
        self.num_replicas, self.rank, self.shuffle, self.epoch = num_replicas, rank, shuffle, 0
        self.num_samples = -(-self.n * self.REPEATS // num_replicas)        # ceil(3n / world)
        self.total_size = self.num_samples * num_replicas
        self.num_selected_samples = (self.n // 256 * 256) // num_replicas

    def __iter__(self):
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.epoch))
        else:
            order = torch.arange(self.n)
        stream = order.repeat_interleave(self.REPEATS)
        pad = self.total_size - stream.numel()
        if pad > 0:
            stream = torch.cat([stream, stream[:pad]])
        mine = stream[self.rank:self.total_size:self.num_replicas]
        return iter(mine[:self.num_selected_samples].tolist())

    def __len__(self):
        return self.num_selected_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def all_reduce_f64(values):
    """Sum a short list of Python floats across ranks in float64 (metric totals)."""
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return t.tolist()


def setup_for_distributed(is_master):
    """Silence print on non-master ranks unless force=True (reference utils.py:288-300)."""
    import builtins as __builtin__
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


def init_distributed_mode(args):
    """env:// rendezvous from torchrun / OMPI / SLURM variables (reference utils.py:339-375).
    Backend is RCCL ("nccl" on ROCm) when a GPU is visible, gloo otherwise (CPU tests)."""
    if "OMPI_COMM_WORLD_RANK" in os.environ:
        args.rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
    elif "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(torch.cuda.device_count(), 1)
        args.world_size = int(os.environ.get("SLURM_NTASKS", 1))
    else:
        print("Not using distributed mode")
        args.distributed = False
        args.rank, args.world_size, args.gpu = 0, 1, 0
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, getattr(args, "dist_url", "env://"), args.gpu),
          flush=True)
    kwargs = {}
    if use_gpu:
        kwargs["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank, **kwargs)
    dist.barrier()
    setup_for_distributed(args.rank == 0)
