"""CPU oracle (torch-CPU ops) for the Face-DeId learned-optics camera.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference
Face-DeId/Camera/Optics.py:10-129 (Camera.__init__/get_psf/forward) and
Face-DeId/Camera/Utils.py:7-57 with the same dtypes (everything f32 / c64) and the
same quirks: fftn/ifftn run over ALL THREE dims of the [3,N,N] field
(Optics.py:101,105), shifts are torch.roll by -/+N//2 (Utils.py:15-30).
"""
import numpy as np
import torch


def sellmeier_delta(lb):
    """Utils.py:33-40 deta: |n_lens(lambda_um) - n_air(lambda_um)|."""
    lens = torch.sqrt(1 + (0.6961663 * (lb ** 2) / ((lb ** 2) - 0.0684043 ** 2)
                           + 0.4079426 * (lb ** 2) / ((lb ** 2) - 0.1162414 ** 2)
                           + 0.8974794 * (lb ** 2) / ((lb ** 2) - 9.896161 ** 2)))
    air = 1 + 0.05792105 / (238.0185 - lb ** -2) + 0.00167917 / (57.362 - lb ** -2)
    return torch.abs(lens - air)


def roll_fwd(x, dim):          # Utils.py:24-30 "fftshift": roll by -(n//2)
    dim = dim if isinstance(dim, tuple) else (dim,)
    return torch.roll(x, tuple(-(x.size(d) // 2) for d in dim), dim)


def roll_inv(x, dim):          # Utils.py:15-21 "ifftshift": roll by +(n//2)
    dim = dim if isinstance(dim, tuple) else (dim,)
    return torch.roll(x, tuple(x.size(d) // 2 for d in dim), dim)


def cexp(phase):               # Utils.py:55-57
    return torch.complex(torch.cos(phase), torch.sin(phase))


def constants(n):
    """Optics.py:13-55: all input-independent tensors, device cpu."""
    c = {}
    zi, z0 = 50e-3, 5.0
    f = 1 / (1 / zi + 1 / z0)
    r_ = f * sellmeier_delta(torch.tensor(550e-9 * 1e6))
    radii = 2.0e-3
    pi = torch.tensor([np.pi])
    l_len = 2 * radii * 2
    px = 3.713103e-6
    l_sen = px * n
    lamb = (torch.tensor([640, 550, 440]) * 1.e-9).unsqueeze(-1).unsqueeze(-1)
    flmb = r_ / sellmeier_delta(lamb * 1e6)
    k = 2 * pi / lamb
    du = l_len / n
    u = torch.arange(-1 * l_len / 2, l_len / 2, du)
    x, y = torch.meshgrid(u, u, indexing="ij")
    xy = x * x + y * y
    rad = torch.sqrt(x ** 2 + y ** 2) <= radii
    fx1 = roll_fwd(torch.arange(-1 / (2 * du), 1 / (2 * du), 1 / l_len), (0,))
    fxx, fyy = torch.meshgrid(fx1, fx1, indexing="ij")
    ff = fxx * fxx + fyy * fyy
    dx2 = l_sen / n
    x2 = torch.arange(-1 * l_sen / 2, l_sen / 2, dx2)
    x2x, x2y = torch.meshgrid(x2, x2, indexing="ij")
    xy2 = x2x * x2x + x2y * x2y
    rho = (torch.sqrt(x2x ** 2 + x2y ** 2) > px * 32) * 1.
    c.update(zi=zi, pi=pi, L_len=l_len, L_sen=l_sen, lamb=lamb, flmb=flmb, k=k, du=du, dx2=dx2,
             XY=xy, rad=rad, FF=ff, XY2=xy2, rho=rho, z=torch.tensor([0.75]), N=n)
    return c


def get_psf(c, coeffs, volume):
    """Optics.py:92-120.  coeffs [K,1,1] f32 (concatenated), volume [K,N,N] f32.
    Returns (psfs [1,3,N,N] f32, loss_rad)."""
    k, flmb, pi, lamb = c["k"], c["flmb"], c["pi"], c["lamb"]
    zi, l_len, l_sen = c["zi"], c["L_len"], c["L_sen"]
    hmap = torch.sum(coeffs * volume, dim=0).unsqueeze(0)                    # Optics.py:79-83
    phase_shift = k * flmb * hmap                                            # Optics.py:89-90
    psfs, loss_rad = None, None
    for dis in c["z"]:
        t = cexp(-(k / (2 * flmb)) * c["XY"])
        focus = cexp((k / (2 * dis)) * c["XY"])
        ph = torch.mul(c["rad"], torch.mul(t, focus)) * cexp(phase_shift)
        vu = torch.mul(ph, cexp((pi / (lamb * zi * l_len) * (l_len - l_sen)) * c["XY"]))
        vu = torch.fft.fftn(roll_fwd(vu, (-2, -1)))                          # all 3 dims!
        vu = torch.mul(vu, cexp(-(pi * lamb * zi * l_len / l_sen) * c["FF"]))
        vu = roll_inv(torch.fft.ifftn(vu), (-2, -1))                         # all 3 dims!
        vu = (l_sen / l_len) * torch.multiply(
            vu, cexp(-(pi / (lamb * zi * l_sen) * (l_len - l_sen)) * c["XY2"]))
        psf = torch.square(torch.abs(vu * ((c["du"] * c["du"]) / (c["dx2"] * c["dx2"]))))
        psf = psf / torch.sum(psf)
        loss_rad = torch.norm(c["rho"] * psf, 'fro')
        psfs = psf.unsqueeze(0) if psfs is None else torch.cat([psfs, psf.unsqueeze(0)], dim=0)
    return psfs, loss_rad


def conv2d_circular(img, kernel):
    """Utils.py:7-12."""
    return torch.fft.irfftn(torch.fft.rfftn(img, dim=(-2, -1)) * torch.fft.rfftn(kernel, dim=(-2, -1)),
                            dim=(-2, -1))


def forward(c, img, coeffs, volume):
    """Optics.py:122-129.  Returns (img_sensor, psfs, loss_rad, centering_loss)."""
    psf, loss_rad = get_psf(c, coeffs, volume)
    n = c["N"]
    cl = torch.mean(torch.square(psf - torch.roll(psf, shifts=img.size(-2) // 2, dims=-2)))
    cl = cl + torch.mean(torch.square(psf - torch.roll(psf, shifts=img.size(-1) // 2, dims=-1)))
    rolled = torch.roll(psf, shifts=(-(n // 2), -(n // 2)), dims=(-2, -1))
    out = conv2d_circular(img, rolled)
    out = torch.div(out, out.amax((1, 2, 3))[:, None, None, None])
    return out, psf, loss_rad, cl
