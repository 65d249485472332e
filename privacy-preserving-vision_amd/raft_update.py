"""Drop-in for reference ``Face-DeId/RAFT/core/update.py:33 SepConvGRU`` (RAFT's separable convolutional GRU) on MI355X.

Same constructor, parameter names (``convz1 .. convq2`` with biases) and ``forward(h, x) -> h`` on NCHW f32 tensors.  RAFT is a
frozen flow estimator in the reference (inference, iterated 12-32 times per pair), so the step runs under an internal
``no_grad``.  Every convolution is the hand-written MFMA implicit GEMM (``ppv_conv_gemm_rect``: 1 x 5 and 5 x 1 taps) in the
fp32-accurate three-term bf16 split of ``ppv_amd.fan`` (the recurrence feeds its own output back 12+ times: plain bf16 products
would accumulate 2^-9 per step); z and r share their input and run as ONE 256-column GEMM; the gates are two fused HIP kernels."""
import torch
from torch import nn

from . import _lib, convops as co
from ._lib import check, ptr, stream_ptr


def _w3(w):
    """conv weight f32 [Cout,Cin,R,S] -> bf16 GEMM rows over the [hi | lo | hi] split input: [W_hi | W_hi | W_lo]"""
    w = w.detach().float()
    hi = w.bfloat16().float()
    lo = (w - hi).bfloat16().float()
    w3 = torch.cat([hi, hi, lo], dim=1)
    assert w3.shape[1] % 64 == 0
    return co.weight_layout(w3.contiguous(), 0)


def _split3(t):
    B, H, W, C = t.shape
    y = torch.empty((B, H, W, 3 * C), dtype=torch.bfloat16, device=t.device)
    check(_lib.lib().ppv_bn_act_split3(ptr(t), None, ptr(y), B * H * W, C, 3 * C, 0, C, stream_ptr()), "ppv_bn_act_split3")
    return y


def _conv(t3, w3, R, S, ph, pw):
    B, H, W, C3 = t3.shape
    N = w3.shape[0]
    out = torch.empty((B, H, W, N), dtype=torch.float32, device=t3.device)
    check(_lib.lib().ppv_conv_gemm_rect(ptr(t3), ptr(w3), ptr(out), ptr(co.zero_page(t3.device)), B, H, W, C3, N, R, S, ph, pw, 1,
                                        stream_ptr()), "ppv_conv_gemm_rect")
    return out


class SepConvGRU(nn.Module):
    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self.convz1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convr1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convq1 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (1, 5), padding=(0, 2))
        self.convz2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convr2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self.convq2 = nn.Conv2d(hidden_dim + input_dim, hidden_dim, (5, 1), padding=(2, 0))
        self._cache = None

    def _weights(self):
        key = tuple((p._version, p.data_ptr()) for p in self.parameters())
        if self._cache is None or self._cache[0] != key:
            c = []
            for z, r, q in ((self.convz1, self.convr1, self.convq1), (self.convz2, self.convr2, self.convq2)):
                c.append((_w3(torch.cat([z.weight, r.weight], 0)), torch.cat([z.bias, r.bias]).detach().float().contiguous(),
                          _w3(q.weight), q.bias.detach().float().contiguous()))
            self._cache = (key, c)
        return self._cache[1]

    @torch.no_grad()
    def forward(self, h, x):
        if not h.is_cuda:
            raise RuntimeError("ppv_amd SepConvGRU runs on an MI355X (cuda tensors); no CPU path")
        Ch = h.shape[1]
        if Ch % 4 or (Ch + x.shape[1]) * 3 % 64:
            raise ValueError("ppv_amd SepConvGRU: hidden_dim % 4 == 0 and 3 (hidden_dim + input_dim) % 64 == 0 (MFMA GEMM tiles)")
        L = _lib.lib()
        hh = h.float().permute(0, 2, 3, 1).contiguous()                    # NHWC
        xx = x.float().permute(0, 2, 3, 1).contiguous()
        B, H, W, _ = hh.shape
        rows = B * H * W
        for (wzr, bzr, wq, bq), (R, S, ph, pw) in zip(self._weights(), ((1, 5, 0, 2), (5, 1, 2, 0))):
            zr = _conv(_split3(torch.cat([hh, xx], dim=3)), wzr, R, S, ph, pw)                 # [.., 2 Ch]: z | r pre-activations
            z, rh = torch.empty_like(hh), torch.empty_like(hh)
            check(L.ppv_gru_zr(ptr(zr), zr.shape[3], ptr(bzr), ptr(hh), ptr(z), ptr(rh), rows, Ch, stream_ptr()), "ppv_gru_zr")
            q = _conv(_split3(torch.cat([rh, xx], dim=3)), wq, R, S, ph, pw)
            hn = torch.empty_like(hh)
            check(L.ppv_gru_out(ptr(q), q.shape[3], ptr(bq), ptr(z), ptr(hh), ptr(hn), rows, Ch, stream_ptr()), "ppv_gru_out")
            hh = hn
        return hh.permute(0, 3, 1, 2).contiguous().to(h.dtype)
