"""torch-facing wrappers of the FFT image (x) PSF convolution kernels (csrc/fftconv.hip)."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _workspace(B, C, N, device):
    nbytes = _lib.lib().ppv_fftconv_workspace_bytes(B, C, N)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def otf_build(psf_chw_view, P, N, workspace=None):
    """psf_chw_view: any strided view indexable as [C, P, P] (f32 or f64) -> OTF^T [C, N/2+1, N] c64."""
    _lib.require_cuda(psf_chw_view)
    C = psf_chw_view.shape[0]
    assert psf_chw_view.shape[1] == P and psf_chw_view.shape[2] == P
    assert psf_chw_view.dtype in (torch.float32, torch.float64)
    dev = psf_chw_view.device
    ws = workspace if workspace is not None else _workspace(1, C, N, dev)
    otf = torch.empty((C, N // 2 + 1, N), dtype=torch.complex64, device=dev)
    sc, sy, sx = psf_chw_view.stride()
    check(_lib.lib().ppv_otf_build(ptr(psf_chw_view), int(psf_chw_view.dtype == torch.float64), sc, sy, sx,
                                   C, P, N, ptr(otf), ptr(ws), stream_ptr()), "ppv_otf_build")
    return otf


def fftconv_fwd(img, otf, mode, conj_otf=False, workspace=None):
    """mode 0 (IC): img [B,C,P,P] -> (|conv| [B,C,P,P], signs, partial_max);  mode 1 (FD): circular, [B,C,N,N].
    img: float32, or uint8 pixels that the row transform decodes as x / 255 (datasets.py:46)."""
    _lib.require_cuda(img, otf)
    img = img.contiguous()
    B, C, H, W = img.shape
    N = 2 * H if mode == 0 else H
    assert H == W and otf.shape == (C, N // 2 + 1, N) and img.dtype in (torch.float32, torch.uint8)
    dev = img.device
    ws = workspace if workspace is not None else _workspace(B, C, N, dev)
    out = torch.empty(img.shape, dtype=torch.float32, device=dev)
    ppi = _lib.lib().ppv_fftconv_partials_per_image(C, N, mode)
    partial = torch.empty(B * ppi, dtype=torch.float32, device=dev)
    signs = torch.zeros(B * C * H * (N // 128), dtype=torch.int64, device=dev) if mode == 0 else None   # row P-1 is never written
    fn = _lib.lib().ppv_fftconv_fwd_u8 if img.dtype == torch.uint8 else _lib.lib().ppv_fftconv_fwd
    check(fn(ptr(img), ptr(otf), ptr(out), ptr(signs), ptr(partial), ptr(ws), B, C, N, mode, int(conj_otf), stream_ptr()), "ppv_fftconv_fwd")
    return out, signs, partial


def group_max(partial, groups):
    out = torch.empty(groups, dtype=torch.float32, device=partial.device)
    check(_lib.lib().ppv_group_max(ptr(partial), ptr(out), groups, partial.numel() // groups, stream_ptr()),
          "ppv_group_max")
    return out


def div_by_group_(x, m):
    groups = m.numel()
    check(_lib.lib().ppv_div_by_group(ptr(x), ptr(m), x.numel() // groups, groups, stream_ptr()), "ppv_div_by_group")
    return x
