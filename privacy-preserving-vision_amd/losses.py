"""The harness' camera MSE term on the MI355X -- ``loss_cam = 1 - criterion_camera(imgs, sensor)`` with
``criterion_camera = nn.MSELoss()`` (Image_Caption/train.py:170-171, 284-288) -- as two fused launches (csrc/loss.hip).

* ``mse_loss(a, b)``          : value and gradients of ``torch.nn.functional.mse_loss(a, b)`` (mean reduction), one pass forward, one
  pass backward; deterministic (partials summed in a fixed order).
* ``one_minus_mse(imgs, s)``  : the expression of train.py:287 as one call.
* ``camera_mse_tap(imgs, s)`` : ``(s', 1 - mse(imgs, s))`` where ``s'`` is ``s`` (same storage) routed THROUGH the loss node: feed ``s'`` to the
  encoder and the gradient the encoder sends back is added to the loss' own gradient inside the loss' backward kernel -- autograd's
  separate ``add_`` pass over the [B,3,H,W] tensor (and ``mse_loss_backward``'s own write) disappears.  Same numbers as the three torch
  ops; 4 reads + 1 write of the batch per step instead of 8 + 3.

No CPU path: the tensors live on the GPU and the HIP library does the work (a missing library raises in ``_lib``)."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_WS = {}


def _workspace(dev):
    """Per-device, per-stream partial-sum workspace of ppv_mse_fwd: zeroed once here, left ready by every call."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _WS:
        _WS[key] = torch.zeros(int(_lib.lib().ppv_mse_workspace_bytes()), dtype=torch.uint8, device=dev)
    return _WS[key]


def _prep(a, b):
    if not (a.is_cuda and b.is_cuda):
        raise RuntimeError("ppv_amd.losses runs on an MI355X (cuda tensors); no CPU path")
    if a.shape != b.shape:
        raise ValueError(f"mse: shapes differ: {tuple(a.shape)} vs {tuple(b.shape)} (the reference's nn.MSELoss would broadcast with a warning)")
    return a.contiguous().float(), b.contiguous().float()


def _fwd(a, b):
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(_lib.lib().ppv_mse_fwd(ptr(a), ptr(b), a.numel(), ptr(_workspace(a.device)), ptr(out), stream_ptr()), "ppv_mse_fwd")
    return out


def _bwd(g_in, a, b, g, coef, want_a):
    g = g.contiguous().float()
    g_b = torch.empty_like(b)
    g_a = torch.empty_like(a) if want_a else None
    gi = None
    if g_in is not None:
        gi = g_in.contiguous().float()
    check(_lib.lib().ppv_mse_bwd(ptr(gi), ptr(a), ptr(b), ptr(g), coef, ptr(g_b), ptr(g_a), a.numel(), stream_ptr()), "ppv_mse_bwd")
    return g_a, g_b


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, sign):
        a, b = _prep(a, b)
        ctx.save_for_backward(a, b)
        ctx.sign = sign
        m = _fwd(a, b)
        return m if sign > 0 else 1.0 - m

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        if not ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]:
            return None, None, None
        g_a, g_b = _bwd(None, a, b, g, ctx.sign * 2.0 / a.numel(), ctx.needs_input_grad[0])
        return g_a, (g_b if ctx.needs_input_grad[1] else None), None


class _TapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, imgs, sensor):
        a, b = _prep(imgs, sensor)
        ctx.save_for_backward(a, b)
        ctx.set_materialize_grads(False)
        return sensor.view(sensor.shape), 1.0 - _fwd(a, b)

    @staticmethod
    def backward(ctx, g_sensor, g_loss):
        a, b = ctx.saved_tensors
        if g_loss is None:                                       # the loss term is unused: pure pass-through
            return None, g_sensor
        g_a, g_b = _bwd(g_sensor, a, b, g_loss, -2.0 / a.numel(), ctx.needs_input_grad[0])
        return g_a, g_b


def mse_loss(input, target):
    """torch.nn.functional.mse_loss(input, target) (reduction='mean')."""
    return _MseFn.apply(input, target, 1)


def one_minus_mse(imgs, sensor):
    """train.py:287: ``1 - criterion_camera(imgs, sensor)``."""
    return _MseFn.apply(imgs, sensor, -1)


def camera_mse_tap(imgs, sensor):
    """-> (sensor routed through the loss node, 1 - mse(imgs, sensor)).  Use the returned tensor as the encoder's input."""
    return _TapFn.apply(imgs, sensor)


class MSELoss(torch.nn.Module):
    """nn.MSELoss() (mean reduction), train.py:171."""

    def forward(self, input, target):
        return mse_loss(input, target)
