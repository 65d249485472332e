"""Data-parallel gradient averaging for the Camera + Encoder step: one process per GPU,
``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on MI355X; ``gloo`` in CPU tests), gradients all-reduced on a
side HIP stream while the rest of backward runs.  The reference has no distributed code; this is the build's own layer.

Two ways in:

* ``attach(params)`` (what ``ppv_amd.encoder.Encoder`` does when a ``GradSync`` is assigned to ``encoder.grad_sync``): the
  gradients of ``params`` -- listed in the order backward produces them -- live in persistent, PRE-FLATTENED f32 buckets.
  Backward's kernels write straight into a parameter's slice (``grad_view``), the trunk sets ``param.grad`` to that slice
  itself and hands autograd nothing, and a bucket is all-reduced in place the moment its last slice is ``mark_ready``:
  no ``torch.cat`` / ``copy_`` round trip (2 x 170 MB per step before), and no tensor that is being all-reduced ever goes
  through autograd's AccumulateGrad (which may add to or clone it on the main stream while the side stream reduces it).
* ``push(grad)``: loose tensors (tests, foreign modules); bucketed by bytes, flattened with one ``cat`` per bucket.

Stream contract: every all-reduce runs on ``self.stream`` behind the events of the gradients it covers.  The trunk's backward
ends with ``join()`` (current stream waits for the side stream), so ``param.grad`` is final for whatever is enqueued after
``backward()`` -- safe by construction.  A harness that wants the tail bucket to overlap what follows backward (bench.py: the
camera's own backward) sets ``defer_join = True`` and calls ``flush()`` before the optimiser step."""
import torch
import torch.distributed as dist


_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev = getattr(torch._C, "_cuda_getDevice", None)


def _raw_current_stream():
    """Handle of the current stream of the current device (0.3 us; torch.cuda.current_stream() builds a Stream object: ~8 us)."""
    if _raw is not None and _dev is not None:
        return _raw(_dev())
    return torch.cuda.current_stream().cuda_stream


class GradSync:
    def __init__(self, bucket_mb=32, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.stream = None
        self.defer_join = False
        self.launched = 0
        # average inside the collective where the backend has it (RCCL: ncclAvg, no separate scaling pass over every bucket)
        try:
            self._avg = dist.get_backend(process_group) == "nccl"
        except Exception:
            self._avg = False
        # loose tensors
        self._bucket, self._size, self._events = [], 0, []
        # pre-flattened buckets
        self._flat = []            # one f32 tensor per bucket
        self._slot = {}            # id(param) -> (bucket, offset, numel)
        self._pending = []         # per bucket: slices not yet marked ready this step
        self._count = []           # per bucket: number of slices
        self._bstreams = []        # per bucket: the streams that wrote slices this step (raw handle -> Stream); one event each at completion
        self._views = {}           # id(param) -> its slice of the flat bucket, shaped like the parameter (built once per layout)
        self._cur_raw, self._cur_stream = None, None
        self._params = []

    # ------------------------------------------------------------------ pre-flattened buckets
    def attach(self, params):
        """params: the parameters whose gradients this object owns, in the order backward finishes them."""
        params = [p for p in params if p.requires_grad]
        if not params:                                                # nothing trainable (fine_tune(False)): no buckets
            self._params, self._flat, self._slot, self._count, self._pending, self._bstreams, self._views = [], [], {}, [], [], [], {}
            return
        if [id(p) for p in params] == [id(p) for p in self._params] and self._flat and self._flat[0].device == params[0].device:
            # same layout as last step.  The per-bucket counters are reset HERE, at the start of every bucketed backward: a backward
            # that aborted midway (an exception caught and retried by the harness) leaves them partly decremented, and the next step
            # would then fire a bucket's all-reduce before all its slices are written (r2 advisor)
            self._pending = list(self._count)
            self._bstreams = [{} for _ in self._count]
            return
        self._params = params
        self._flat, self._slot, self._count = [], {}, []
        cur, off = [], 0
        groups = []
        for p in params:
            n = p.numel()
            cur.append((p, off, n))
            off += (n + 3) // 4 * 4                                   # 16-byte aligned slices
            if off * 4 >= self.bucket_bytes:
                groups.append((cur, off))
                cur, off = [], 0
        if cur:
            groups.append((cur, off))
        for b, (items, total) in enumerate(groups):
            self._flat.append(torch.zeros(total, dtype=torch.float32, device=params[0].device))
            self._count.append(len(items))
            for p, o, n in items:
                self._slot[id(p)] = (b, o, n)
        self._pending = [c for c in self._count]
        self._bstreams = [{} for _ in self._count]
        self._views = {}
        for p in params:
            b, o, n = self._slot[id(p)]
            self._views[id(p)] = self._flat[b][o:o + n].view(p.shape)

    def grad_view(self, p):
        """The slice of the flat bucket that IS this parameter's gradient (None: not attached).  The same tensor object every step."""
        return self._views.get(id(p))

    def _events_of(self, b):
        """One event per stream that wrote into bucket b this step, recorded NOW (after the last write was enqueued).  (Rounds 1-2
        created and recorded an event per PARAMETER: 312 per step, ~1.2 ms of host time.)"""
        streams, self._bstreams[b] = self._bstreams[b], {}
        events = []
        for st in streams.values():
            ev = torch.cuda.Event()
            ev.record(st)
            events.append(ev)
        return events

    def mark_ready(self, p, stream=None):
        """The slice of ``p`` has been written on ``stream`` (default: the CURRENT stream); the bucket is reduced when all its slices are."""
        b = self._slot[id(p)][0]
        if self._flat[b].is_cuda:
            if stream is None:
                raw = _raw_current_stream()
                if raw != self._cur_raw or self._cur_stream is None:
                    self._cur_raw, self._cur_stream = raw, torch.cuda.current_stream()
                stream, key = self._cur_stream, raw
            else:
                key = id(stream)
            self._bstreams[b][key] = stream
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._pending[b] = self._count[b]
            self._reduce_on_side(self._flat[b], self._events_of(b), [])

    # ------------------------------------------------------------------ loose tensors
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
        if len(grads) == 1:
            self._reduce_on_side(grads[0].reshape(-1), events, grads)
            return
        self._reduce_on_side(None, events, grads)

    def _reduce_on_side(self, flat, events, scatter_to):
        self.launched += 1
        ref = flat if flat is not None else scatter_to[0]

        def run():
            buf = flat if flat is not None else torch.cat([g.reshape(-1) for g in scatter_to])
            self._all_reduce_mean(buf)
            if flat is None:
                off = 0
                for g in scatter_to:
                    g.copy_(buf[off:off + g.numel()].view_as(g))
                    off += g.numel()

        if ref.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=ref.device)
            for ev in events:                        # every gradient of this bucket, on its producing stream
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                run()
                for g in scatter_to:
                    g.record_stream(self.stream)
        else:
            run()

    def _all_reduce_mean(self, buf):
        if self._avg:
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
        else:                                        # gloo (CPU tests, one-GPU rehearsals) has no AVG
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                buf.mul_(1.0 / self.world)

    # ------------------------------------------------------------------ joins
    def launch_pending(self):
        """Start the all-reduce of the last, partly filled bucket of loose tensors."""
        self._launch()

    def join(self):
        """Current stream waits for every all-reduce issued so far."""
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def end_of_backward(self):
        """Called by the trunk when it has no more gradients: tail bucket out; join unless the harness deferred it."""
        self._launch()
        for b, c in enumerate(self._count):          # a bucket some of whose slices never came (frozen mid-way): reduce what there is
            if self._pending[b] != c:
                self._pending[b] = c
                self._reduce_on_side(self._flat[b], self._events_of(b), [])
        if not self.defer_join:
            self.join()

    def flush(self):
        """Launch what is left and make the current stream wait for every outstanding all-reduce.  With ``defer_join`` this MUST
        run between ``backward()`` and the optimiser step (bench.py does)."""
        self._launch()
        self.join()

    def reduce_now(self, tensors):
        """Blocking-order average of a few small tensors (lens coefficients) on the current stream."""
        for t in tensors:
            if t is not None:
                self._all_reduce_mean(t)
