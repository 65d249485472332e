"""Data-parallel gradient averaging for the Camera + Encoder step: one process per GPU,
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on MI355X; ``gloo`` in CPU tests), gradients bucketed in the
order backward produces them (reverse layer order) and all-reduced on a side HIP stream while the rest of backward
runs.  The reference has no distributed code; this is the build's own layer."""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, bucket_mb=32, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.stream = None
        self._bucket, self._size, self._events = [], 0, []
        self.launched = 0

    # -- called by Encoder backward as each gradient tensor becomes final
    def push(self, grad):
        if grad is None:
            return
        self._bucket.append(grad)
        if grad.is_cuda:                         # the gradient is complete on whichever stream produced it
            ev = torch.cuda.Event()
            ev.record()
            self._events.append(ev)
        self._size += grad.numel() * grad.element_size()
        if self._size >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        grads, self._bucket, self._size = self._bucket, [], 0
        events, self._events = self._events, []
        if not grads:
            return
        self.launched += 1
        if grads[0].is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=grads[0].device)
            for ev in events:                        # every gradient of this bucket, on its producing stream
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self._reduce(grads)
                for g in grads:
                    g.record_stream(self.stream)
        else:
            self._reduce(grads)

    def _reduce(self, grads):
        flat = torch.cat([g.reshape(-1) for g in grads]) if len(grads) > 1 else grads[0].reshape(-1)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        if len(grads) > 1:
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()

    def launch_pending(self):
        """Start the all-reduce of the last, partly filled bucket (called when the producer has no more gradients)."""
        self._launch()

    def flush(self):
        """Launch what is left and make the current stream wait for every outstanding all-reduce.  MUST run between
        ``backward()`` and the optimiser step (bench.py does; a training script calls ``encoder.grad_sync.flush()``)."""
        self._launch()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def reduce_now(self, tensors):
        """Blocking-order average of a few small tensors (lens coefficients) on the current stream."""
        for t in tensors:
            if t is not None:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.mul_(1.0 / self.world)
