"""``ppv_amd.optim.Adam``: torch.optim.Adam with the whole parameter list updated by ONE hand-written kernel launch.

Reference use: the harness' optimisers, ``Image_Caption/train.py:92-101`` (``torch.optim.Adam(params=filter(lambda p: p.requires_grad,
encoder.parameters()), lr=encoder_lr)`` and the decoder's twin), stepped at ``train.py:318-321``.  Same constructor arguments, same
``param_groups`` / ``state`` layout (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter, so ``state_dict()`` loads into
``torch.optim.Adam`` and back), same arithmetic as torch's fused kernel in f32 (``csrc/optim.hip``).  Not supported and refused loudly:
``amsgrad``, ``maximize``, ``capturable``, sparse or non-f32 gradients.  No CPU / stock-torch fallback: the HIP library does the step."""
import struct

import numpy as np
import torch

from . import _lib

_CHUNK = 4096


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False, capturable=False,
                 **ignored):
        if amsgrad or maximize or capturable:
            raise NotImplementedError("ppv_amd.optim.Adam: amsgrad / maximize / capturable are not implemented")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("ppv_amd.optim.Adam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None))
        self._tables = {}                      # group index -> (key of data pointers, device descriptor table, total blocks)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)    # new exp_avg / exp_avg_sq tensors: every cached descriptor table is stale
        self._tables = {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._tables = {}

    def _table(self, gi, ps):
        # the key holds EVERY pointer the records hold (parameter, gradient, both moments): a reloaded state or a re-allocated gradient
        # rebuilds the table instead of updating freed buffers
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        recs, blk = [], 0
        for p in ps:
            st = self.state[p]
            ptrs = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr())
            vec = int(all(q % 16 == 0 for q in ptrs))
            recs.append(struct.pack("<QQQQqii", *ptrs, p.numel(), blk, vec))
            blk += (p.numel() + _CHUNK - 1) // _CHUNK
        desc = torch.from_numpy(np.frombuffer(b"".join(recs), dtype=np.uint8).copy()).to(ps[0].device)
        self._tables[gi] = (key, desc, blk)
        return desc, blk

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            for p in ps:
                g = p.grad
                if (p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse or not p.is_contiguous() or not g.is_contiguous()
                        or p.device != dev or dev.type != "cuda"):
                    raise NotImplementedError("ppv_amd.optim.Adam: contiguous f32 parameters and gradients on one GPU only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            by_step = {}                       # parameters whose gradient was missing in some step keep their own count (as torch's do)
            for p in ps:
                by_step.setdefault(float(self.state[p]["step"]), []).append(p)
            b1, b2 = group["betas"]
            if len(by_step) > 1:               # split case: tables keyed by (group, slot), slots reused every step -> no growth with t
                for k in [k for k in self._tables if isinstance(k, tuple) and k[0] == gi and k[1] >= len(by_step)]:
                    del self._tables[k]
            for slot, (t, sub) in enumerate(sorted(by_step.items())):
                desc, blocks = self._table((gi, slot) if len(by_step) > 1 else gi, sub)
                n = t + 1.0
                _lib.check(L.ppv_adam_multi(desc.data_ptr(), len(sub), blocks, group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                            1.0 - b1 ** n, 1.0 - b2 ** n, _lib.stream_ptr()), "ppv_adam_multi")
                for p in sub:
                    self.state[p]["step"] += 1
        return loss
