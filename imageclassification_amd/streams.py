"""Second HIP stream for work that hangs off the backward chain (weight gradients).

A layer's weight gradient needs only that layer's output gradient and its saved input, and nothing before the optimizer
needs its result, so it can run beside the main stream's data-gradient / normalisation-backward chain: the wgrad GEMMs are
MFMA- and latency-bound, the chain's LayerNorm / GELU / BatchNorm passes are HBM-bound, and together they fill both.
`SideLane` keeps the bookkeeping: an event orders each side launch after the main-stream producer of its inputs, and a
per-buffer event stops the main stream from overwriting a buffer a pending side launch still reads."""
import torch


class SideLane:
    def __init__(self, device, enabled=True):
        self.enabled = bool(enabled)
        self.side = torch.cuda.Stream(device=device) if self.enabled else None
        self._pending = {}    # data_ptr -> event of the last side launch reading that buffer
        self._last = None

    def begin(self):
        self.main = torch.cuda.current_stream()
        self._pending.clear()
        self._last = None

    @property
    def stream_ptr(self):
        return self.side.cuda_stream if self.enabled else self.main.cuda_stream

    def launch(self, fn, reads=()):
        """fn(stream_ptr) enqueues the work; `reads` are data_ptr()s of buffers the main stream will overwrite later."""
        if not self.enabled:
            fn(self.main.cuda_stream)
            return
        ready = torch.cuda.Event()
        ready.record(self.main)
        self.side.wait_event(ready)
        fn(self.side.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.side)
        self._last = done
        for p in reads:
            self._pending[p] = done

    def before_write(self, *ptrs):
        """Main stream is about to overwrite these buffers: wait for side launches that still read them."""
        for p in ptrs:
            ev = self._pending.pop(p, None)
            if ev is not None:
                self.main.wait_event(ev)

    def events(self):
        """Events a consumer on another stream (the gradient all-reduce) must wait for besides the main stream."""
        return () if self._last is None else (self._last,)

    def join(self):
        """Main stream waits for everything launched on the side lane so far (before a gradient hook / the optimizer)."""
        if self._last is not None:
            self.main.wait_event(self._last)
            self._last = None
        self._pending.clear()
