"""CPU check behind the c64 Fresnel transform of round 5 (csrc/psf_ic.hip, PPV_PSF_F32): the oracle camera (oracle/ic_camera.py = the
reference's Lens.py / Utils.py restated) with its two FFTs of propagate_fresnel in complex64 instead of complex128 -- PSF, sensor image
and lens gradient against the c128 run.  The reference's fields are c64-VALUED (plate and spherical wavefront are cast to c64 before
they are multiplied, Utils.py:80-85; the transfer function is c64) and only typed c128 by the f64 aperture mask.
Measured (R = 448, K = 36, P = 128): sensor 4.8e-7, psf 4.1e-7, gradient 5.1e-7 (max-norm relative)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import ic_camera as ic
from oracle import zernike as oz

torch.set_num_threads(8)
R, K, P = 448, 36, 128
vol = torch.tensor(oz.zernike_volume(R, K), dtype=torch.float32)
c = torch.zeros(K, 1, 1)
c[3] = -11.0
c[4:] += (torch.rand(K - 4, 1, 1, generator=torch.Generator().manual_seed(1)) - 0.5) * 0.4
img = torch.rand(2, 3, P, P, generator=torch.Generator().manual_seed(0))
noise = torch.rand(1, R, R, 1, generator=torch.Generator().manual_seed(1))
w = torch.rand(2, 3, P, P, generator=torch.Generator().manual_seed(5))


def run(c64):
    orig = ic.propagate_fresnel
    if c64:
        def pf(field, distance, sample_interval, wave_lengths):
            _, m_orig, n_orig, _ = field.shape
            mp, np_ = m_orig // 4, n_orig // 4
            padded = F.pad(field.to(torch.complex64), [0, 0, np_, np_, mp, mp])
            h = ic.fresnel_transfer(m_orig, n_orig, sample_interval, wave_lengths, distance)
            obj = torch.fft.fftn(padded.permute(0, 3, 1, 2), dim=[-1, -2]).permute(0, 2, 3, 1)
            out = torch.fft.ifftn((obj * h).permute(0, 3, 1, 2), dim=[-1, -2]).permute(0, 2, 3, 1)
            return out[:, mp:-mp, np_:-np_, :]
        ic.propagate_fresnel = pf
    cc = c.clone().requires_grad_(True)
    s, psf, _ = ic.forward(img, cc, vol, noise, prueba=None, height_tolerance=2e-8, sensor_distance=0.025, sample_interval=3e-6)
    (s * w).sum().backward()
    ic.propagate_fresnel = orig
    return s.detach(), psf.detach(), cc.grad.detach()


a, b = run(False), run(True)
for n, x, y in zip(("sensor", "psf", "lens gradient"), a, b):
    print(f"{n:14s} c64 vs c128 transform: max-norm relative difference {float((x.double() - y.double()).abs().max() / x.double().abs().max()):.2e}")
