"""CPU oracle (torch-CPU ops) for the Image_Caption learned-optics camera.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, stage by stage and with the
same dtype promotions, reference Image_Caption/Camera/Lens.py:141-318
(OpticsZernike.forward) and the functions of Image_Caption/Camera/Utils.py it calls.
All functions are differentiable through torch autograd, so the same code is the
gradient oracle.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_WAVE_LENGTHS = np.array([460, 550, 640]) * 1e-9      # Lens.py:18
DEFAULT_REFRACTIVE_IDCS = np.array([1.499, 1.493, 1.488])    # Lens.py:17


def cexp64_to_c64(phase):
    """Utils.py:80-85 compl_exp_tf: phase -> f64, cos/sin in f64, each cast to c64."""
    phase = phase.to(torch.float64)
    return torch.cos(phase).to(torch.complex64) + 1j * torch.sin(phase).to(torch.complex64)


def height_map(coeffs, volume):
    """Lens.py:176-177: sum_k c_k Z_k -> [1,R,R,1] f32."""
    return torch.sum(coeffs * volume, dim=0).unsqueeze(0).unsqueeze(-1)


def phase_plate(hmap, noise_u01, wave_lengths, refractive_idcs, height_tolerance):
    """Utils.py:396-410 + :192-205.  ``noise_u01`` is the U[0,1) draw of Utils.py:403 (f32, hmap shape)."""
    if height_tolerance is not None:
        hmap = hmap + ((-height_tolerance - height_tolerance) * noise_u01 + height_tolerance)
    delta_n = refractive_idcs.reshape([1, 1, 1, -1]) - 1.0
    wave_nos = (2.0 * np.pi / wave_lengths).reshape([1, 1, 1, -1])
    phi = torch.tensor(wave_nos * delta_n) * hmap          # f64 * f32 -> f64
    return cexp64_to_c64(phi)


def spherical_wavefront(wave_res, physical_size, wave_lengths, depth=0.5):
    """Lens.py:191-210 (optics_cfg == 1 -> depth 1/2)."""
    n, m = wave_res
    x, y = np.mgrid[-n // 2:n // 2, -m // 2:m // 2].astype(np.float64)
    x = x / n * physical_size
    y = y / m * physical_size
    sq = x ** 2 + y ** 2
    wave_nos = torch.tensor((2.0 * np.pi / wave_lengths).reshape([1, 1, 1, -1]))
    curv = torch.sqrt(torch.tensor(sq) + torch.tensor(depth, dtype=torch.float64) ** 2)
    curv = curv.unsqueeze(0).unsqueeze(-1)
    return cexp64_to_c64(wave_nos * curv)


def circular_aperture(field):
    """Utils.py:88-97: f64 mask (r < max(x)) * field  => complex128."""
    s = list(field.shape)
    x, y = np.mgrid[-s[1] // 2:s[1] // 2, -s[2] // 2:s[2] // 2].astype(np.float64)
    r = np.sqrt(x ** 2 + y ** 2)[None, :, :, None]
    return torch.tensor((r < np.amax(x)).astype(np.float64)) * field


def fresnel_transfer(m_orig, n_orig, sample_interval, wave_lengths, distance):
    """Utils.py:339-373: H = exp(-i pi lambda z (fx^2+fy^2)) on the ifftshifted padded grid -> c64."""
    mp, np_ = m_orig // 4, n_orig // 4
    m, n = m_orig + 2 * mp, n_orig + 2 * np_
    x, y = np.mgrid[-n // 2:n // 2, -m // 2:m // 2]
    fx = np.fft.ifftshift(x / (sample_interval * n))
    fy = np.fft.ifftshift(y / (sample_interval * m))
    sq = (np.square(fx) + np.square(fy))[None, :, :, None]
    tmp = np.float64(wave_lengths * np.pi * -1.0 * sq * distance)
    return cexp64_to_c64(torch.tensor(tmp, dtype=torch.float64))


def propagate_fresnel(field, distance, sample_interval, wave_lengths):
    """Utils.py:328-378 (NHWC field; FFT over H,W)."""
    _, m_orig, n_orig, _ = field.shape
    mp, np_ = m_orig // 4, n_orig // 4
    padded = F.pad(field, [0, 0, np_, np_, mp, mp])
    h = fresnel_transfer(m_orig, n_orig, sample_interval, wave_lengths, distance)
    obj = torch.fft.fftn(padded.permute(0, 3, 1, 2), dim=[-1, -2]).permute(0, 2, 3, 1)
    out = torch.fft.ifftn((obj * h).permute(0, 3, 1, 2), dim=[-1, -2]).permute(0, 2, 3, 1)
    return out[:, mp:-mp, np_:-np_, :]


def nearest_resize(x, size):
    """torchvision Resize(interpolation=0) on a tensor == legacy nearest: src = floor(dst * in/out)."""
    return F.interpolate(x, size=size, mode="nearest")


def area_downsample(img, target):
    """Utils.py:216-248 (NHWC in, NHWC f32 out)."""
    side = img.shape[1]
    img = img.to(torch.float32).permute(0, 3, 1, 2)
    if side % target == 0:
        f = side // target
        out = F.avg_pool2d(img, f, stride=f)
    else:
        lcm = abs(target * side) / math.gcd(target, side) / target
        up = 10 if lcm > 10 else int(lcm)
        out = F.avg_pool2d(nearest_resize(img, [up * target, up * target]), up, stride=up)
    return out.permute(0, 2, 3, 1)


def otf_from_psf(psfs, out_hw):
    """Utils.py:127-158 psf2otf: psfs [h,w,1,C] -> OTF [out_h,out_w,1,C] c64 (centre lands on index (1,1))."""
    fh, fw = psfs.shape[0], psfs.shape[1]
    padded = psfs
    if out_hw[0] != fh:
        pad = (out_hw[0] - fh) / 2
        if (out_hw[0] - fh) % 2 != 0:
            p0, p1 = int(np.ceil(pad)), int(np.floor(pad))
        else:
            p0, p1 = int(pad) + 1, int(pad) - 1
        padded = F.pad(psfs, [0, 0, 0, 0, p0, p1, p0, p1])
    hh, ww = padded.shape[0], padded.shape[1]

    def perm(n):
        split = n - (n + 1) // 2
        return np.concatenate((np.arange(split, n), np.arange(split)))

    padded = padded[perm(hh)][:, perm(ww)]
    tmp = padded.permute(2, 3, 0, 1)
    return torch.fft.fftn(tmp.to(torch.complex64), dim=[-1, -2]).permute(2, 3, 0, 1)


def img_psf_conv(img, psfs):
    """Utils.py:251-297, circular=False, adjoint=False.  img [B,C,H,W] f32, psfs [h,w,1,C]."""
    h = img.shape[2]
    target = 2 * h
    hp = (target - h) / 2
    pt, pb = int(np.ceil(hp)), int(np.floor(hp))
    padded = F.pad(img, [pt, pb, pt, pb])
    img_fft = torch.fft.fftn(padded, dim=[-1, -2])
    otf = otf_from_psf(psfs, padded.shape[2:]).permute(2, 3, 0, 1)
    res = torch.abs(torch.fft.ifftn(img_fft * otf, dim=[-1, -2]))
    res = res[:, :, pt + 1:-pb, pt + 1:-pb]
    return nearest_resize(res, list(img.shape[2:]))


def disk_masks(size=256, radius=32):
    """Lens.py:111-127: mask_1 = 1 outside the disk, mask_2 = 1 inside; HWC f64."""
    from .zernike import filled_disk
    d = filled_disk(size, (size // 2, size // 2), radius)
    m1 = np.repeat((~d).astype(np.float64)[:, :, None], 3, axis=2)
    m2 = np.repeat(d.astype(np.float64)[:, :, None], 3, axis=2)
    return torch.from_numpy(m1), torch.from_numpy(m2)


def forward(img, coeffs, volume, noise_u01, *, prueba=None, mask_1=None, mask_2=None,
            wave_lengths=DEFAULT_WAVE_LENGTHS, refractive_idcs=DEFAULT_REFRACTIVE_IDCS,
            height_tolerance=20e-9, sensor_distance=25e-3, sample_interval=2e-6,
            patch_size=None, return_intermediates=False):
    """OpticsZernike.forward (Lens.py:141-318), upsample=False, psf_lab=None.

    img [B,3,P,P] f32; coeffs [K,1,1] f32 (already concatenated); volume [K,R,R] f32;
    noise_u01 [1,R,R,1] f32 = the torch.rand draw of Utils.py:403.
    Returns (sensor, psf, loss) (+ dict of intermediates)."""
    res = volume.shape[-1]
    patch_size = patch_size or img.shape[-1]
    physical_size = float(res * sample_interval)
    hmap = height_map(coeffs, volume)
    plate = phase_plate(hmap, noise_u01, wave_lengths, refractive_idcs, height_tolerance)
    sph = spherical_wavefront((res, res), physical_size, wave_lengths)
    field = circular_aperture(sph * plate)
    sensor_field = propagate_fresnel(field, sensor_distance, sample_interval, wave_lengths)
    intensity = torch.square(torch.abs(sensor_field))                       # Utils.py:208-209
    psf = area_downsample(intensity, patch_size)                             # Lens.py:238
    psf = psf / torch.sum(psf, dim=[1, 2], keepdim=True)                     # Lens.py:239
    psf_n = psf
    loss = None
    if prueba in ("1", "3"):
        loss = torch.norm((psf * mask_1) - psf)                              # Lens.py:271
    if prueba in ("2", "3"):
        psf = psf * mask_2                                                   # Lens.py:274
    raw = img_psf_conv(img, psf.permute(1, 2, 0, 3))                         # Lens.py:280,290
    sensor = raw / raw.max()                                                 # Lens.py:312
    if return_intermediates:
        return sensor, psf, loss, dict(height_map=hmap, field=field, sensor_field=sensor_field,
                                       intensity=intensity, psf_normalised=psf_n, raw=raw)
    return sensor, psf, loss
