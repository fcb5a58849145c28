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
import torch
import torch.distributed as dist


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
    """Bucketed, overlapped all-reduce of a flat gradient tensor. Works on CUDA (RCCL, side stream) and on CPU
    tensors (gloo; used by the world_size-2 tests)."""

    def __init__(self, flat_grad, first_bucket_mb=1.0, bucket_mb=25.0, process_group=None, last_bucket_mb=4.0):
        self.flat = flat_grad
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        esz = flat_grad.element_size()
        self.buckets = make_buckets(flat_grad.numel(), int(first_bucket_mb * (1 << 20)) // esz,
                                    int(bucket_mb * (1 << 20)) // esz, last_bucket_elems=int(last_bucket_mb * (1 << 20)) // esz)
        self.on_gpu = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        self._next = 0
        self._works = []
        self.launched = []   # bucket indices in launch order (tests)

    def reset(self):
        self._next = 0
        self._works = []
        self.launched = []

    def grads_ready_from(self, lo, wait_events=()):
        """Backward reports that every gradient at offset >= lo is final once the current stream and `wait_events`
        (events of other streams that also write gradients: the weight-gradient side lane) have been reached."""
        if self.world == 1:
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
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Flush remaining buckets and make the compute stream wait for all reductions."""
        if self.world == 1:
            return
        self.grads_ready_from(0)
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
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
                 last_bucket_mb=4.0):
        self.module = module
        self.reducer = GradReducer(module.grad_arena, first_bucket_mb, bucket_mb, last_bucket_mb=last_bucket_mb)
        module.grad_ready_hook = lambda lo, _hi=None, events=(): self.reducer.grads_ready_from(lo, events)
        # DDP constructor semantics: every rank starts from rank 0's parameters and buffers
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(module.param_arena, src=0)
            dist.broadcast(module.buffer_arena, src=0)
            module.refresh_shadow()

    def sync_buffers(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.module.buffer_arena, src=0)

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
