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


def ic_transform_length(P):
    """Transform length of the IC sensor convolution for patch size P: 2 P where that is 256 / 512 (the kernels' native sizes), else
    the next of 256 / 512 / 1024 >= 2 P (supports P x P: any length >= 2 P - 1 gives the reference's 2 P-point convolution)."""
    for n in (256, 512, 1024):
        if 2 * P <= n:
            return n
    raise ValueError("patch sizes above 512 are outside the image convolution kernels (1024-point transforms)")


def fftconv_ic_fwd(img, otf, N, return_workspace=False):
    """IC sensor convolution (Utils.py:251-297), img [B,C,P,P] f32 or uint8, any even P with 2 P <= N in {256, 512, 1024}
    -> (|conv| [B,C,P,P], signs, partial_max[, workspace]).  The workspace starts with the row transform of the image
    (B*C*P*(N/2) float2), which ppv_fftconv_ic_bwd_p can take back instead of recomputing it."""
    _lib.require_cuda(img, otf)
    img = img.contiguous()
    B, C, P, W = img.shape
    assert P == W and P % 2 == 0 and 2 * P <= N and otf.shape == (C, N // 2 + 1, N) and img.dtype in (torch.float32, torch.uint8)
    dev = img.device
    L = _lib.lib()
    ws = torch.empty(L.ppv_fftconv_ic_workspace_bytes(B, C, P, N), dtype=torch.uint8, device=dev)
    out = torch.empty(img.shape, dtype=torch.float32, device=dev)
    partial = torch.empty(L.ppv_fftconv_ic_partials(B, C, P), dtype=torch.float32, device=dev)
    signs = torch.zeros(B * C * P * (N // 128), dtype=torch.int64, device=dev)          # row P-1 is never written
    check(L.ppv_fftconv_ic_fwd_p(ptr(img), int(img.dtype == torch.uint8), ptr(otf), ptr(out), ptr(signs), ptr(partial), ptr(ws),
                                 B, C, P, N, stream_ptr()), "ppv_fftconv_ic_fwd_p")
    return (out, signs, partial, ws) if return_workspace else (out, signs, partial)


def group_max(partial, groups):
    out = torch.empty(groups, dtype=torch.float32, device=partial.device)
    check(_lib.lib().ppv_group_max(ptr(partial), ptr(out), groups, partial.numel() // groups, stream_ptr()),
          "ppv_group_max")
    return out


def div_by_group_(x, m):
    groups = m.numel()
    check(_lib.lib().ppv_div_by_group(ptr(x), ptr(m), x.numel() // groups, groups, stream_ptr()), "ppv_div_by_group")
    return x
