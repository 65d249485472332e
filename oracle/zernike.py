"""Noll-ordered, Noll-normalised Zernike basis on a square pixel grid.

Restates the published algorithm of ``poppy.zernike.zernike_basis(nterms, npix,
outside=0.0)`` (poppy 1.0.3, pinned in reference Image_Caption/environment.yml:154),
which the reference calls at Image_Caption/Camera/Utils.py:75-77 and
Face-DeId/Camera/Utils.py:60-63.  poppy is not vendored in /root/reference, so this
is *parity unpinned* against poppy itself; analytic checks live in
tests/test_zernike.py.

Conventions (poppy): j = 1..nterms Noll index; even j -> cos(m theta), odd j ->
sin(m theta); normalisation sqrt(n+1) (x sqrt(2) for m != 0); grid
x_i = (i - (npix-1)/2) / ((npix-1)/2), xx varies along the last axis; rho > 1 -> outside.
"""
from math import factorial

import numpy as np


def noll_indices(j):
    """Noll index j (1-based) -> (n, m); sign of m: even j positive (cosine)."""
    if j < 1:
        raise ValueError("Noll indices start at 1")
    n = 0
    j1 = j - 1
    while j1 > n:
        n += 1
        j1 -= n
    m = (-1) ** j * ((n % 2) + 2 * int((j1 + ((n + 1) % 2)) / 2.0))
    return n, m


def radial_coeffs(n, m):
    """[(coef, power)] of the radial polynomial R_n^|m|(rho) (explicit factorial sum)."""
    m = abs(m)
    out = []
    if (n - m) % 2:
        return out
    for k in range((n - m) // 2 + 1):
        coef = ((-1) ** k * factorial(n - k)
                / (factorial(k) * factorial((n + m) // 2 - k) * factorial((n - m) // 2 - k)))
        out.append((coef, n - 2 * k))
    return out


def zernike_basis(nterms, npix, outside=0.0):
    """float64 [nterms, npix, npix]."""
    x = (np.arange(npix, dtype=np.float64) - (npix - 1) / 2.0) / ((npix - 1) / 2.0)
    xx, yy = np.meshgrid(x, x)
    rho = np.sqrt(xx ** 2 + yy ** 2)
    theta = np.arctan2(yy, xx)
    inside = rho <= 1.0
    nmax = noll_indices(nterms)[0]
    pw = [np.ones_like(rho)]
    for _ in range(nmax):
        pw.append(pw[-1] * rho)
    out = np.empty((nterms, npix, npix), dtype=np.float64)
    trig = {}
    for j in range(1, nterms + 1):
        n, m = noll_indices(j)
        rad = np.zeros_like(rho)
        for coef, p in radial_coeffs(n, m):
            rad += coef * pw[p]
        if m == 0:
            z = rad * np.sqrt(n + 1.0) if n else np.ones_like(rho)
        else:
            key = (abs(m), m > 0)
            if key not in trig:
                trig[key] = np.cos(abs(m) * theta) if m > 0 else np.sin(abs(m) * theta)
            z = (np.sqrt(2.0) * np.sqrt(n + 1.0)) * rad * trig[key]
        out[j - 1] = np.where(inside, z, outside)
    return out


def zernike_volume(resolution, n_terms, scale_factor=1e-6):
    """Reference get_zernike_volume (IC Utils.py:75-77, FD Utils.py:60-63): basis * 1e-6 (float64)."""
    return zernike_basis(n_terms, resolution, outside=0.0) * scale_factor


def filled_disk(size, center, radius):
    """Euclidean filled disk d^2 <= r^2 (stand-in for cv2.circle(..., thickness=-1), IC Lens.py:111-118).

    *Parity unpinned* against OpenCV's rasteriser (boundary pixels may differ)."""
    yy, xx = np.mgrid[0:size, 0:size]
    return ((xx - center[0]) ** 2 + (yy - center[1]) ** 2) <= radius * radius
