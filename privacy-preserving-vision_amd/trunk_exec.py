"""Host side of csrc/trunk_plan.hip: the whole trunk of ``ppv_amd.encoder.Encoder`` (Image_Caption/models.py:31-41 under train-mode
BatchNorm, train.py:245) from ONE FFI call per direction over one persistent arena.

The per-kernel path of encoder.py crosses ctypes ~200 times per step, takes ~460 tensors from the caching allocator and zeroes two pools
with their own launches; it stays as the reference form (tests' taps, per-class timing, eval mode, the opt-in schedules).  This module
is what the default train-mode step runs: the same launches in the same order (csrc/trunk_plan.hip calls the same entry points with the
same arguments), no allocation, one memset.

Ownership.  A ``_Lease`` = (arena, flat gradient buffer, pointer tables).  Forward takes a free lease (or makes one), the autograd
context holds it, and it returns to the pool when the context dies (after backward, or at once under ``no_grad``); the next forward on
the same arena is ordered behind whatever still reads it by an event recorded at release.  Gradients of the trainable parameters live
in the lease's flat f32 buffer: backward's kernels write a parameter's slice and the trunk returns fresh views of the slices to
autograd, which adopts them as ``param.grad`` without a copy (as DDP's gradient_as_bucket_view: the next backward overwrites them) --
unless a gradient is already there (micro-batching, ``zero_grad(set_to_none=False)``): then a temporary buffer is used and autograd
accumulates.  Data-parallel runs (``dist_sync.GradSync``) keep setting ``param.grad`` to the bucket slices themselves."""
import ctypes
import os
import weakref

import torch

from . import _lib
from . import convops as co

BF16, F32 = torch.bfloat16, torch.float32
_UNSUPPORTED = object()                    # marks a geometry in Encoder._plans that TrunkPlan refused


_NF = len(_lib.TRUNK_CONV_FIELDS)          # pointers per convolution record
_WT, _WD, _GAMMA, _BETA, _RM, _RV, _DW, _DG, _DB = range(_NF)


class _Lease:
    __slots__ = ("arena", "gflat", "table", "hyper", "ptr_key", "grad_items", "free_event", "plan", "gen", "handed", "__weakref__")


class _LeaseHolder:
    """Kept by the autograd context.  The lease goes back to its plan's pool at the end of backward (or when the context dies without
    one: ``no_grad``, an abandoned graph); a second backward through the same context (``retain_graph=True``) finds the arena intact as
    long as no forward has taken the lease in between (generation stamp), else it is refused."""
    __slots__ = ("lease", "gen", "held")

    def __init__(self, lease):
        self.lease, self.gen, self.held = lease, lease.gen, True

    def release(self):
        if self.held:
            self.held = False
            lease = self.lease
            try:
                with torch.cuda.device(lease.arena.device):
                    if torch.cuda.is_current_stream_capturing():
                        ev = None                     # a captured step is ordered by its own streams; an event of the capture must not leak out
                    else:
                        ev = torch.cuda.Event()
                        ev.record()
                lease.free_event = ev
            except Exception:                         # interpreter shutdown
                lease.free_event = None
            plan = lease.plan
            enc = plan.owner() if plan.owner is not None else None      # weak: the plan must not keep its encoder (or its plan table) alive
            if plan.owner is not None and (enc is None or enc.__dict__.get("_plans", {}).get(plan.key) is not plan):
                return                    # the plan has been replaced (other requires_grad pattern / parameter list) or its encoder is gone: the arena is not pooled
            plan.free.append(lease)

    def retake(self):
        """For a repeated backward: the lease again, if nothing has overwritten what forward kept."""
        if self.held:
            return self.lease
        lease = self.lease
        if lease.gen != self.gen:
            raise RuntimeError("ppv_amd Encoder: backward through a graph whose saved activations have been overwritten by a later "
                               "forward (the trunk keeps them in a reused arena); run this backward before the next forward")
        if lease in lease.plan.free:           # (not there: the pool dropped its idle arenas for another geometry -- this one is intact)
            lease.plan.free.remove(lease)
        lease.gen += 1                         # whoever else still points at this lease is stale from here on
        self.gen = lease.gen
        self.held = True
        return lease

    def __del__(self):
        self.release()


class TrunkPlan:
    """Geometry-specific state: the descriptor csrc/trunk_plan.hip derives the arena layout from, the flat-gradient layout, leases."""

    def __init__(self, enc, B, H, W, fold_rows, reduce3, rg_key):
        self.recs = [enc._stem] + [r for blk in enc._blocks for r in blk]        # None where a block has no projection
        self.nblocks = len(enc._blocks)
        if self.nblocks > _lib.TRUNK_MAX_BLOCKS:
            raise ValueError("trunk too deep for the plan executor")
        d = _lib.TrunkDesc()
        d.B, d.H, d.W, d.nblocks, d.fold_rows, d.wgrad_reduce3 = B, H, W, self.nblocks, fold_rows, int(reduce3)
        for i, (r1, r2, r3, rd) in enumerate(enc._blocks):
            k = d.blk[i]
            k.planes, k.stride, k.proj = r1.conv.out_channels, r2.stride, int(rd is not None)
            k.train_w = sum(bit for bit, r in ((1, r1), (2, r2), (4, r3), (8, rd)) if r is not None and r.conv.weight.requires_grad)
        self.desc = d
        self.bytes = _lib.lib().ppv_trunk_arena_bytes(ctypes.byref(d))
        if self.bytes == 0:
            raise ValueError("ppv_trunk_arena_bytes: geometry not supported by the plan executor")
        self.c_last = 4 * enc._blocks[-1][0].conv.out_channels
        h, w = H // 4, W // 4
        for (_, r2, _, _) in enc._blocks:
            h, w = h // r2.stride, w // r2.stride
        self.hw_last = (h, w)
        # flat gradient layout: every trainable tensor a 16-byte aligned slice, in record order (conv weight, BN weight, BN bias)
        self.slots = []                                # (record index, field, parameter, offset, numel)
        off = 0
        for i, r in enumerate(self.recs):
            if r is None:
                continue
            for field, p in ((_DW, r.conv.weight), (_DG, r.bn.weight), (_DB, r.bn.bias)):
                if p.requires_grad:
                    self.slots.append((i, field, p, off, p.numel()))
                    off += (p.numel() + 3) // 4 * 4
        self.grad_elems = max(off, 4)
        # every slice a multiple of 4 floats (true for every ResNet shape): the buffer is densely packed and ONE C++ call cuts it into
        # parameter-shaped views (torch._utils._unflatten_dense_tensors) instead of two Python-level ops per parameter
        self.dense = all(sl[4] % 4 == 0 for sl in self.slots)
        self.grad_params = [sl[2] for sl in self.slots]
        # trainable parameters of each block in the order the per-kernel backward finishes them (the bucket order of data-parallel runs)
        self.block_params = []
        for blk in enc._blocks:
            self.block_params.append([p for r in (blk[2], blk[1], blk[0], blk[3]) if r is not None
                                      for p in (r.bn.weight, r.bn.bias, r.conv.weight) if p.requires_grad])
        self.stem_trainable = any(p.requires_grad for p in (enc._stem.conv.weight, enc._stem.bn.weight, enc._stem.bn.bias))
        self.free = []
        self.rg_key = rg_key
        self.key, self.owner = None, None

    # ------------------------------------------------------------------ leases
    def acquire(self, dev):
        if self.free:
            lease = self.free.pop()
            if lease.arena.device == dev:
                if lease.free_event is not None:
                    if not torch.cuda.is_current_stream_capturing():      # (waiting for uncaptured work is an error inside a capture)
                        torch.cuda.current_stream(dev).wait_event(lease.free_event)
                    lease.free_event = None
                lease.gen += 1
                return lease
        lease = _Lease()
        lease.gen = 0
        lease.plan = self
        lease.arena = torch.empty(self.bytes, dtype=torch.uint8, device=dev)
        lease.gflat = torch.zeros(self.grad_elems, dtype=F32, device=dev)
        n = len(self.recs)
        lease.table = (ctypes.c_uint64 * (_NF * n))()
        lease.hyper = (ctypes.c_float * (2 * n))()
        lease.ptr_key = None
        lease.free_event = None
        lease.handed = ()                  # weak references to the views of gflat the last backward gave to autograd
        base = lease.gflat.data_ptr()
        lease.grad_items = [(sl[2], None) for sl in self.slots]
        for i, field, p, off, numel in self.slots:
            lease.table[_NF * i + field] = base + 4 * off
        return lease

    def grad_views(self, flat):
        """{parameter: fresh view of its slice of `flat`} -- fresh objects, so that autograd's AccumulateGrad adopts them as .grad
        without a copy (it clones a gradient somebody else still references)."""
        if not self.slots:
            return {}
        if self.dense:
            n = self.slots[-1][3] + self.slots[-1][4]
            return dict(zip(self.grad_params, torch._utils._unflatten_dense_tensors(flat[:n], self.grad_params)))
        return {p: flat[off:off + numel].view(p.shape) for _, _, p, off, numel in self.slots}

    def fill_pointers(self, enc, lease, tok, plist):
        """Weight-layout / BatchNorm columns of the lease's table.  Refilled only when something they depend on changed: the parameter
        list object (Encoder._param_list: rebuilt on .to() / load_state_dict(assign=True) / replaced Parameters), the shared bf16
        layouts of the trainable convolutions, or version / storage of a frozen weight."""
        frozen = tuple((r.conv.weight._version, r.conv.weight.data_ptr()) for r in self.recs
                       if r is not None and not r.conv.weight.requires_grad)
        wl = getattr(enc, "_wl", None)
        frozen = (enc.__dict__.get("_wcache_gen", 0),) + frozen          # invalidate_weight_cache() drops the cached layouts themselves
        # the BatchNorm storages themselves (bn.running_mean = t, p.data = t, a parent's load_state_dict(assign=True) replace them
        # without touching the parameter list): ~400 integer compares per step against kernels writing through a stale pointer
        bnp = tuple(t.data_ptr() if t is not None else 0 for r in self.recs if r is not None
                    for t in (r.bn.weight, r.bn.bias, r.bn.running_mean, r.bn.running_var))
        k = lease.ptr_key
        if k is not None and k[0] is plist and k[1] is wl and k[2] == frozen and k[3] == bnp:
            return
        t, hy = lease.table, lease.hyper
        for i, r in enumerate(self.recs):
            if r is None:
                continue
            bn = r.bn
            b = _NF * i
            t[b + _WT], t[b + _WD] = r.wt(tok).data_ptr(), r.wd(tok).data_ptr()
            t[b + _GAMMA], t[b + _BETA] = bn.weight.data_ptr(), bn.bias.data_ptr()
            t[b + _RM] = bn.running_mean.data_ptr() if bn.running_mean is not None else 0
            t[b + _RV] = bn.running_var.data_ptr() if bn.running_var is not None else 0
            hy[2 * i], hy[2 * i + 1] = bn.momentum, bn.eps
        lease.ptr_key = (plist, wl, frozen, bnp)


def usable(enc, train):
    """Whether the default train-mode step of this encoder can go through the plan executor (else: the per-kernel path)."""
    if not train or os.environ.get("PPV_TRUNK_PLAN", "1") == "0" or os.environ.get("PPV_BLOCK_EXEC", "1") == "0":
        return False
    env = os.environ.get
    if env("PPV_BN_FOLD_ACT", "1") != "1" or env("PPV_BN_FUSED", "0") == "1" or env("PPV_BN_FOLD_THREAD", "0") == "1":
        return False
    if int(env("PPV_WGRAD_GROUP", "0")) or env("PPV_WGRAD_SCHED", "024") != "024" or int(env("PPV_DGRAD_BNRED", "3")) != 3:
        return False
    if env("PPV_STEM_BWD_FUSED", "1") == "0":
        return False
    if co.PROFILE is not None:
        return False
    if enc._stem.conv.weight.requires_grad or (enc.grad_sync is not None and enc._stem.bn.weight.requires_grad):
        return False
    for r in [enc._stem] + [r for blk in enc._blocks for r in blk if r is not None]:
        if not r.bn.training or r.bn.momentum is None or not r.bn.affine:
            return False
    if len(enc._blocks) > _lib.TRUNK_MAX_BLOCKS:           # deeper than the executor's descriptor: the per-kernel path serves it
        return False
    return True


def get_plan(enc, B, H, W, plist):
    fold_rows = max(1, int(os.environ.get("PPV_BN_FOLD_ROWS", "2")))
    if torch.are_deterministic_algorithms_enabled():
        fold_rows = 32                     # 32 partial rows: <= 2 adders per address up to 8192 pixels (bit-reproducible there), fewest possible beyond
    reduce3 = os.environ.get("PPV_WGRAD_REDUCE3", "0") == "1"
    rg = tuple(p.requires_grad for p in plist)
    key = (B, H, W, fold_rows, reduce3)
    plans = enc.__dict__.setdefault("_plans", {})
    plan = plans.get(key)
    if plan is _UNSUPPORTED:
        return None
    if plan is None or plan.rg_key != rg or plan.plist is not plist:
        for other in plans.values():         # a new geometry: idle arenas of the others (19.6 GB each at B = 128) go back to the allocator
            if other is not _UNSUPPORTED:
                other.free.clear()
        if plans.get(key) is _UNSUPPORTED:
            return None
        try:
            plan = TrunkPlan(enc, B, H, W, fold_rows, reduce3, rg)
        except ValueError:                # a geometry the executor's layout refuses: remembered, the per-kernel path runs instead
            plans[key] = _UNSUPPORTED
            return None
        plan.plist = plist                # the list object itself (Encoder._param_list rebuilds it when Parameter objects are replaced)
        plan.key, plan.owner = key, weakref.ref(enc)
        plans[key] = plan
    return plan


class _BlockTap(tuple):
    """What the tests' stage-wise checks read of a block the per-kernel path would have saved: [0] block input, [11] block output,
    [12] sign mask of the block input (None for the first block) -- views of the arena, valid until the next forward on it."""


def block_taps(plan, lease, cells, stem=None):
    """Per-block views into the arena (tests only; Encoder._debug_block_grads): [( xin, ..., yout, xin_bits )] in block order."""
    L = _lib.lib()
    d, B = plan.desc, plan.desc.B
    off = (ctypes.c_size_t * 20)()
    _lib.check(L.ppv_trunk_block_offsets(ctypes.byref(d), -1, off), "ppv_trunk_block_offsets")
    arena = lease.arena

    def view(o, shape, dt=BF16):
        n = 1
        for k in shape:
            n *= k
        return arena[o:o + n * (2 if dt == BF16 else 1)].view(dt).view(shape)

    h, w, cin = d.H // 4, d.W // 4, 64
    x = view(off[2], (B, h, w, 64))
    if stem is not None:                   # (raw conv output, BatchNorm coefficients, pooled activation, arg-max) as the per-kernel path keeps them
        stem.append((view(off[0], (B, d.H // 2, d.W // 2, 64)), arena[off[1]:off[1] + 4 * 64 * 4].view(F32).view(4, 64), x,
                     view(off[3], (B, h, w, 64), torch.uint8)))
    bits = None
    out = []
    for i in range(plan.nblocks):
        k = d.blk[i]
        _lib.check(L.ppv_trunk_block_offsets(ctypes.byref(d), i, off), "ppv_trunk_block_offsets")
        h2, w2, c3 = h // k.stride, w // k.stride, 4 * k.planes
        yout = cells if i == plan.nblocks - 1 else view(off[6], (B, h2, w2, c3))
        gin = view(off[19], (B, h, w, cin))
        rec = [None] * 13
        rec[0], rec[11], rec[12] = x, yout, bits
        out.append((_BlockTap(rec), gin))
        x, bits = yout, view(off[7], (B * h2 * w2 * c3 // 8,), torch.uint8)
        h, w, cin = h2, w2, c3
    return out


def forward(enc, images, tok, plist):
    """-> (cells [B,h,w,C] bf16, holder).  images [B,3,H,W] f32 contiguous on the device."""
    B, _, H, W = images.shape
    dev = images.device
    plan = get_plan(enc, B, H, W, plist)
    if plan is None:
        return None, None                  # the executor does not serve this geometry: the caller takes the per-kernel path
    lease = plan.acquire(dev)
    plan.fill_pointers(enc, lease, tok, plist)
    cells = torch.empty((B, plan.hw_last[0], plan.hw_last[1], plan.c_last), dtype=BF16, device=dev)
    _lib.check(_lib.lib().ppv_trunk_fwd(ctypes.byref(plan.desc), lease.table, lease.hyper, images.data_ptr(), lease.arena.data_ptr(),
                                        cells.data_ptr(), co.zero_page(dev).data_ptr(), _lib.stream_ptr()), "ppv_trunk_fwd")
    return cells, _LeaseHolder(lease)


def backward(enc, holder, cells, g_out, g_cells, needs_img, img_shape, side, main_masked=None):
    """-> (g_img or None, {param: gradient} or None).  The dict is None when the trunk set ``param.grad`` itself.
    main_masked: optional stream (a CU-masked one) the main chain of backward runs on instead of the current stream: forked from the
    current stream at the start, joined back at the end."""
    lease = holder.retake()
    plan = lease.plan
    dev = cells.device
    L = _lib.lib()
    zp = co.zero_page(dev).data_ptr()
    cur_ptr = main = _lib.stream_ptr()
    side_ptr = side.cuda_stream if side is not None else None
    hop = main_masked is not None and enc.grad_sync is None
    # gradient of the output
    if g_out is not None and g_cells is not None:
        g = co.adaptive_pool_bwd(g_out.contiguous(), plan.hw_last, relu_of=cells)
        g.add_(co.adaptive_pool_bwd(g_cells.contiguous(), plan.hw_last, relu_of=cells))
        g_top, kind, E = g, 0, 0
    else:
        g_top = (g_out if g_out is not None else g_cells).contiguous()
        if g_top.dtype not in (F32, BF16):
            g_top = g_top.float()
        kind, E = (1 if g_top.dtype == F32 else 2), g_top.shape[1]
    g_img = torch.empty(img_shape, dtype=F32, device=dev) if needs_img else None
    sync = enc.grad_sync
    trainable = [it[0] for it in lease.grad_items]
    fresh = all(p.grad is None for p in trainable)
    table = lease.table
    grads = None
    keep = None
    if sync is not None and fresh and trainable:
        # data-parallel: the kernels write the parameters' slices of the flat gradient buckets; one call per run of blocks that completes
        # a bucket, the bucket's all-reduce starts behind events recorded on both streams right after it
        sync.attach([p for blk in reversed(plan.block_params) for p in blk])
        table = (ctypes.c_uint64 * len(lease.table))(*lease.table)
        for i, field, p, off, numel in plan.slots:
            table[_NF * i + field] = sync.grad_view(p).data_ptr()
        hi = plan.nblocks
        pend = {}
        for b in range(plan.nblocks - 1, -1, -1):
            done = False
            for p in plan.block_params[b]:
                bk = sync._slot[id(p)][0]
                pend[bk] = pend.get(bk, sync._count[bk]) - 1
                done = done or pend[bk] == 0
            if done or b == 0:
                _lib.check(L.ppv_trunk_bwd(ctypes.byref(plan.desc), table, lease.arena.data_ptr(), cells.data_ptr(), g_top.data_ptr(), kind, E,
                                           g_img.data_ptr() if g_img is not None else None, zp, b, hi, main, side_ptr), "ppv_trunk_bwd")
                cur = torch.cuda.current_stream(dev)
                for bb in range(hi - 1, b - 1, -1):
                    for p in plan.block_params[bb]:
                        p.grad = sync.grad_view(p)
                        if side is not None:
                            sync._bstreams[sync._slot[id(p)][0]][id(side)] = side
                        sync.mark_ready(p, stream=cur)
                hi = b
        holder.release()
        return g_img, None
    flat = lease.gflat
    # views of gflat handed out by an earlier backward that somebody still holds and that are NOT the parameters' current .grad (those
    # make `fresh` false): two torch.autograd.grad calls over the parameters, gradients saved before zero_grad(set_to_none=True), ...
    # Writing gflat again would silently replace their contents (plain autograd returns independent tensors): temporary buffer then.
    aliased = any(w() is not None for w in lease.handed)
    if not fresh or sync is not None or aliased:
        # gradients already sit in .grad (micro-batching, zero_grad(set_to_none=False)) -- possibly views of this very buffer from the
        # last step: a temporary flat buffer, whose slices autograd adds to what is there
        flat = keep = torch.empty(plan.grad_elems, dtype=F32, device=dev)
        table = (ctypes.c_uint64 * len(lease.table))(*lease.table)
        for i, field, p, off, numel in plan.slots:
            table[_NF * i + field] = keep.data_ptr() + 4 * off
        if side is not None:
            keep.record_stream(side)
    if hop:
        main = ctypes.c_void_p(main_masked.cuda_stream)
        _lib.check(L.ppv_stream_fork(cur_ptr, main), "ppv_stream_fork")
    _lib.check(L.ppv_trunk_bwd(ctypes.byref(plan.desc), table, lease.arena.data_ptr(), cells.data_ptr(), g_top.data_ptr(), kind, E,
                               g_img.data_ptr() if g_img is not None else None, zp, 0, plan.nblocks, main, side_ptr), "ppv_trunk_bwd")
    if hop:
        _lib.check(L.ppv_stream_fork(main, cur_ptr), "ppv_stream_fork")
    taps = getattr(enc, "_debug_block_grads", None)          # tests: per-block (g_out, g_in), last block first -- as the per-kernel path
    if taps is not None:
        bt = block_taps(plan, lease, cells)
        off = (ctypes.c_size_t * 20)()
        _lib.check(L.ppv_trunk_block_offsets(ctypes.byref(plan.desc), -1, off), "ppv_trunk_block_offsets")
        n_top = cells.numel() * 2
        g_last = g_top if kind == 0 else lease.arena[off[5]:off[5] + n_top].view(BF16).view(cells.shape)
        for i in range(plan.nblocks - 1, -1, -1):
            taps.append((g_last if i == plan.nblocks - 1 else bt[i + 1][1], bt[i][1]))
    if keep is not None and sync is not None:
        sync.reduce_now([keep])            # accumulation mode of a data-parallel run: averaged in stream order, then handed to autograd
    # Single-process runs hand the slices to AUTOGRAD (hooks on the parameters fire, torch.autograd.grad leaves .grad alone, accumulation
    # is autograd's): param.grad ends up a view of the lease's flat buffer, which the next backward overwrites.
    grads = plan.grad_views(flat)
    if keep is None:
        lease.handed = tuple(weakref.ref(t) for t in grads.values())
    holder.release()
    return g_img, grads
