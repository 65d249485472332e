"""Zernike volume built on the GPU (csrc/zernike.hip).  Product-side counterpart of the reference's
``get_zernike_volume`` (Image_Caption/Camera/Utils.py:75-77, Face-DeId/Camera/Utils.py:60-63)."""
import math
import struct

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _noll(j):
    n, j1 = 0, j - 1
    while j1 > n:
        n += 1
        j1 -= n
    m = (-1) ** j * ((n % 2) + 2 * int((j1 + ((n + 1) % 2)) / 2.0))
    return n, m


def _term_table(n_terms):
    recs, coefs = [], []
    for j in range(1, n_terms + 1):
        n, m = _noll(j)
        am = abs(m)
        off = len(coefs)
        for k in range((n - am) // 2 + 1):
            coefs.append((-1) ** k * math.factorial(n - k)
                         / (math.factorial(k) * math.factorial((n + am) // 2 - k) * math.factorial((n - am) // 2 - k)))
        norm = math.sqrt(n + 1.0) if m == 0 else math.sqrt(2.0) * math.sqrt(n + 1.0)
        if n == 0:
            norm = 1.0
        recs.append(struct.pack("<iiiid", n, m, off, len(coefs) - off, norm))
    return np.frombuffer(b"".join(recs), dtype=np.uint8).copy(), np.asarray(coefs, dtype=np.float64)


def zernike_volume(resolution, n_terms, device, scale_factor=1e-6):
    """[n_terms, resolution, resolution] float32 on `device`; basis * scale_factor, 0 outside the unit disk."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("zernike_volume runs on the GPU (libppv_hip); no CPU path in the product")
    if _noll(n_terms)[0] > _lib.lib().ppv_zernike_max_order():
        raise NotImplementedError("radial order above the kernel's table")
    recs, coefs = _term_table(n_terms)
    d_recs = torch.from_numpy(recs).to(device)
    d_coefs = torch.from_numpy(coefs).to(device)
    out = torch.empty((n_terms, resolution, resolution), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(_lib.lib().ppv_zernike_basis(ptr(d_recs), ptr(d_coefs), ptr(out), n_terms, resolution,
                                           float(scale_factor), 0.0, stream_ptr()), "ppv_zernike_basis")
    return out
