"""Data-parallel gradient averaging over RCCL/xGMI (reference: torch DistributedDataParallel wrap at
/root/reference/train.py:218-222; collective inventory SURVEY.md 2.3).

One process per GPU.  Gradients live in ONE flat fp32 arena whose order is the forward layer order, so they
become final from the END of the arena towards the start while backward runs.  The reducer cuts the arena
into buckets from the end (first bucket small so communication starts early, like DDP's 1 MiB first bucket,
then 25 MiB; the bucket holding the front of the arena is capped at 4 MiB because nothing overlaps its all-reduce), and as soon as backward reports that a bucket's whole range is final it enqueues
all_reduce(SUM) for that slice on a side HIP stream (event-chained to the compute stream), overlapping the
remaining backward kernels.  The optimizer kernel applies the 1/world averaging (grad_scale), so no extra
pass over the gradients is needed.  xGMI note: a ring all-reduce is per-link bound; 25 MiB buckets keep each
of the 7 links busy with >3 MiB chunks at 8 GPUs.

BatchNorm buffers: DDP's per-forward broadcast of rank 0's running statistics only matters when the
statistics are READ (evaluate / checkpoint): `sync_buffers()` does that one broadcast on demand.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import hip

DT_F32, DT_I32, DT_F64, DT_BF16 = 0, 1, 2, 3       # include/icamd.h ICAMD_DT_*
RED_SUM, RED_MIN, RED_MAX = 0, 1, 2                # ICAMD_RED_*
_DT = {torch.float32: DT_F32, torch.int32: DT_I32, torch.float64: DT_F64, torch.bfloat16: DT_BF16}


class RcclComm:
    """The library's own RCCL communicator (include/icamd.h `icamd_rccl_*`, csrc/collective.hip): one per process.
    Rank 0 draws the ncclUniqueId; it travels to the other ranks over torch.distributed's control plane (the reference
    sets that process group up in utils.py:339-375).  Collectives are enqueued through the C ABI on a caller-supplied
    HIP stream: `icamd_allreduce_bucket_launch`, `icamd_broadcast_launch`."""

    def __init__(self, rank=None, world=None, group=None):
        self.lib = hip.load()
        if not self.lib.icamd_rccl_available():
            raise hip.IcamdError("librccl could not be loaded: the RCCL transport is unavailable on this host")
        ddp_up = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank(group) if ddp_up else 0) if rank is None else rank
        self.world = (dist.get_world_size(group) if ddp_up else 1) if world is None else world
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            hip.check(self.lib.icamd_rccl_unique_id(buf), "rccl_unique_id")
        blob = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(blob, src=0, group=group)
        handle = ctypes.c_void_p()
        hip.check(self.lib.icamd_rccl_comm_init(blob[0], self.world, self.rank, ctypes.byref(handle)), "rccl_comm_init")
        self.handle = handle

    def info(self):
        """(ranks, rank) as the live communicator reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = ctypes.c_int(0), ctypes.c_int(0)
        hip.check(self.lib.icamd_rccl_comm_info(self.handle, ctypes.byref(n), ctypes.byref(r)), "rccl_comm_info")
        return n.value, r.value

    def all_reduce(self, t, op=RED_SUM, stream=None):
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        hip.check(self.lib.icamd_allreduce_bucket_launch(self.handle, t.data_ptr(), t.numel(), _DT[t.dtype], op, s),
                  "allreduce_bucket_launch")

    def broadcast(self, t, root=0, stream=None):
        s = torch.cuda.current_stream().cuda_stream if stream is None else stream
        hip.check(self.lib.icamd_broadcast_launch(self.handle, t.data_ptr(), t.numel(), _DT[t.dtype], root, s), "broadcast_launch")

    def destroy(self):
        if self.handle is not None:
            self.lib.icamd_rccl_comm_destroy(self.handle)
            self.handle = None


_default_comm = None


def default_comm():
    """Process-wide communicator, created on first use (after init_distributed_mode)."""
    global _default_comm
    if _default_comm is None:
        _default_comm = RcclComm()
    return _default_comm


def destroy_default_comm():
    """ncclCommDestroy of the process-wide communicator (call before dist.destroy_process_group())."""
    global _default_comm
    if _default_comm is not None:
        _default_comm.destroy()
        _default_comm = None


def make_buckets(n_elems, first_bucket_elems, bucket_elems, align=64, last_bucket_elems=0):
    """[(lo, hi)] covering [0, n_elems) from the END backwards; boundaries aligned to `align` elements.

    `last_bucket_elems` > 0 caps the bucket that contains offset 0.  That bucket is launched when backward finishes, so its
    all-reduce is the one nothing overlaps; the front of the arena (stem and first stage) holds few parameters but the
    longest-running backward kernels, so cutting it off lets the rest of its bucket go out while they still run."""
    buckets = []
    hi = n_elems
    cap = first_bucket_elems
    while hi > 0:
        lo = max(0, hi - cap)
        lo = (lo // align) * align
        if lo == 0 and 0 < last_bucket_elems < hi:
            cut = (last_bucket_elems // align) * align
            if cut > 0:
                buckets.append((cut, hi))
                hi = cut
        buckets.append((lo, hi))
        hi = lo
        cap = bucket_elems
    return buckets


class GradReducer:
    """Bucketed, overlapped all-reduce of a flat gradient tensor.

    Transports: "torch" (default) -- torch.distributed's process group: RCCL over xGMI when the backend is nccl, gloo for the
    CPU / shared-GPU world_size-2 tests; "rccl" -- the library's own RCCL call site (`icamd_allreduce_bucket_launch` on the
    side stream through a second communicator), opt-in with transport="rccl" / ICAMD_DDP_TRANSPORT=rccl / `bench.py
    --transport rccl` until a run with two or more GPUs has validated it (tests/test_rccl_gpu.py::test_two_gpu_* is gated on
    torch.cuda.device_count() >= 2).  `force=True` runs the collectives at world size 1 too (the single-GPU test of the real
    transport and of the stream / event chain)."""

    def __init__(self, flat_grad, first_bucket_mb=1.0, bucket_mb=25.0, process_group=None, last_bucket_mb=4.0,
                 transport=None, force=False, comm=None):
        self.flat = flat_grad
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = bool(force)
        self.on_gpu = flat_grad.is_cuda
        if transport is None:
            transport = os.environ.get("ICAMD_DDP_TRANSPORT", "")
        if not transport:
            transport = "torch"
        if transport not in ("rccl", "torch"):
            raise ValueError(f"unknown gradient transport {transport!r}")
        self.transport = transport
        self.comm = None
        if transport == "rccl" and (self.world > 1 or self.force):
            self.comm = comm if comm is not None else default_comm()
        esz = flat_grad.element_size()
        self.buckets = make_buckets(flat_grad.numel(), int(first_bucket_mb * (1 << 20)) // esz,
                                    int(bucket_mb * (1 << 20)) // esz, last_bucket_elems=int(last_bucket_mb * (1 << 20)) // esz)
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self._next = 0
        self._works = []
        self.launched = []   # bucket indices in launch order (tests)
        # called as cb(lo, hi, bucket_index) with the side stream current, right behind that bucket's all-reduce: the
        # per-bucket optimizer launch (engine.train_one_epoch sets it for the steps that can use it).  Device tensors only.
        self.bucket_callback = None
        self.callbacks = []  # bucket indices whose callback ran, in order (tests)
        self._flag_reduced = False

    def reset(self):
        self._next = 0
        self._works = []
        self.launched = []
        self.callbacks = []
        self._flag_reduced = False

    def grads_ready_from(self, lo, wait_events=()):
        """Backward reports that every gradient at offset >= lo is final once the current stream and `wait_events`
        (events of other streams that also write gradients: the weight-gradient side lane) have been reached."""
        if not self.active:
            return
        while self._next < len(self.buckets) and self.buckets[self._next][0] >= lo:
            self._launch(self._next, wait_events)
            self._next += 1

    def _launch(self, bi, wait_events=()):
        lo, hi = self.buckets[bi]
        view = self.flat[lo:hi]
        self.launched.append(bi)
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            for e in wait_events:
                self.comm_stream.wait_event(e)
            if self.comm is not None:
                self.comm.all_reduce(view, RED_SUM, self.comm_stream.cuda_stream)
            else:
                with torch.cuda.stream(self.comm_stream):
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            if self.bucket_callback is not None:
                with torch.cuda.stream(self.comm_stream):
                    self.bucket_callback(lo, hi, bi)
                self.callbacks.append(bi)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_flag(self, finite_flag):
        """MIN-reduce the finite-loss flag (int32 [1]) over the ranks NOW, on the side stream: the flag depends on the
        forward only, so it is final on every rank before the first gradient bucket goes out; everything enqueued on the side
        stream afterwards (the per-bucket optimizer launches) sees the reduced value.  The main stream sees it after
        finish() / wait()."""
        if not self.active:
            return
        self._flag_reduced = True
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())          # the flag is written by the main stream's metrics kernel
            self.comm_stream.wait_event(ev)
            if self.comm is not None:
                self.comm.all_reduce(finite_flag, RED_MIN, self.comm_stream.cuda_stream)
            else:
                with torch.cuda.stream(self.comm_stream):
                    dist.all_reduce(finite_flag, op=dist.ReduceOp.MIN, group=self.group)
        else:
            dist.all_reduce(finite_flag, op=dist.ReduceOp.MIN, group=self.group)

    def wait(self):
        """The main stream waits for everything the side stream holds (no buckets are flushed)."""
        if self.active and self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    @property
    def active(self):
        return self.world > 1 or self.force

    def ranks_seen(self):
        """Ranks in the communicator that carries the gradients (RCCL: asked of the live communicator)."""
        if self.comm is not None:
            return self.comm.info()[0]
        return self.world

    def finish(self, finite_flag=None):
        """Flush remaining buckets and make the compute stream wait for all reductions.

        `finite_flag` (int32 [1], 1 = this rank's loss was finite): reduced with MIN behind the last bucket, so that a
        non-finite loss on ANY rank makes EVERY rank drop the step (the reduced gradients are poisoned for all of them) --
        the reference has no cross-rank agreement here (engine.py:56-59: the bad rank `continue`s and the others hang in
        their all-reduce)."""
        if not self.active:
            return
        self.grads_ready_from(0)
        if self._flag_reduced:
            finite_flag = None           # reduce_flag() already did it for this step
            self._flag_reduced = False
        if self.on_gpu:
            if finite_flag is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())      # the flag is written by the main stream's metrics kernel
                self.comm_stream.wait_event(ev)
                if self.comm is not None:
                    self.comm.all_reduce(finite_flag, RED_MIN, self.comm_stream.cuda_stream)
                else:
                    with torch.cuda.stream(self.comm_stream):
                        dist.all_reduce(finite_flag, op=dist.ReduceOp.MIN, group=self.group)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            if finite_flag is not None:
                self._works.append(dist.all_reduce(finite_flag, op=dist.ReduceOp.MIN, group=self.group, async_op=True))
            for w in self._works:
                w.wait()
        self._works = []
        self._next = 0

    @property
    def grad_scale(self):
        return 1.0 / self.world


class DistributedDataParallel:
    """Wrapper with DDP's surface (`.module`, call-through) for the HIP model."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, first_bucket_mb=1.0, bucket_mb=25.0,
                 last_bucket_mb=4.0, transport=None, force=False):
        self.module = module
        self.reducer = GradReducer(module.grad_arena, first_bucket_mb, bucket_mb, last_bucket_mb=last_bucket_mb,
                                   transport=transport, force=force)
        module.grad_ready_hook = lambda lo, _hi=None, events=(): self.reducer.grads_ready_from(lo, events)
        # DDP constructor semantics: every rank starts from rank 0's parameters and buffers
        if self.reducer.active:
            self._broadcast(module.param_arena)
            self._broadcast(module.buffer_arena)
            module.refresh_shadow()

    def _broadcast(self, t):
        if self.reducer.comm is not None:
            self.reducer.comm.broadcast(t, 0)      # current stream: ordered with the kernels that read the arena next
        elif dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(t, src=0)

    def sync_buffers(self):
        if self.reducer.active:
            self._broadcast(self.module.buffer_arena)

    def shutdown(self):
        """Release the library's own communicator (if this wrapper's reducer made one)."""
        if self.reducer.comm is not None:
            torch.cuda.synchronize()
            if self.reducer.comm is _default_comm:
                destroy_default_comm()
            else:
                self.reducer.comm.destroy()
            self.reducer.comm = None

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        self.module.eval()
        return self

    def parameters(self):
        return self.module.parameters()

    def __call__(self, x):
        return self.module(x)

    def __getattr__(self, name):
        return getattr(self.__dict__["module"], name)
