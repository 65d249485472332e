"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import load_golden, rel_err
from oracle import fd_camera as fd
from oracle import ic_camera as ic
from oracle import zernike as oz

TOL = 1e-3        # north_star: PSF and activations within 1e-3 rel fp32


def test_ic_tiny_forward_and_grad():
    g = load_golden("ic_tiny.npz")
    coeffs = torch.tensor(g["coeffs"], requires_grad=True)
    sensor, psf, loss, inter = ic.forward(
        torch.tensor(g["img"]), coeffs, torch.tensor(g["volume"]), torch.tensor(g["noise_u01"]),
        prueba=None, height_tolerance=2e-8, sensor_distance=0.025, sample_interval=3e-6,
        return_intermediates=True)
    assert loss is None
    assert psf.dtype == torch.float32 and sensor.dtype == torch.float32
    assert rel_err(psf.detach(), g["psf"]) < 1e-5
    assert rel_err(inter["raw"].detach(), g["raw"]) < 1e-5
    assert rel_err(sensor.detach(), g["sensor"]) < 1e-5
    (sensor * torch.tensor(g["w"])).sum().backward()
    got = coeffs.grad.reshape(-1)[3:]
    assert rel_err(got, g["grad_sensor_w"]) < TOL


def test_ic_stage_functions():
    g = load_golden("ic_tiny.npz")
    psfs = torch.tensor(g["psf"]).permute(1, 2, 0, 3)
    otf = ic.otf_from_psf(psfs, [64, 64])
    assert rel_err(otf.real, g["otf"].real) < 1e-6 and rel_err(otf.imag, g["otf"].imag) < 1e-6
    # delta PSF at (16,16): output(i,j) = img(max(i-1,0), max(j-1,0)) -- pins the OTF centre (1,1),
    # the [pad+1:-pad] crop and the nearest 31->32 index map (Utils.py:137-147, :293-295)
    img = torch.tensor(g["img"])
    d = torch.zeros(32, 32, 1, 3)
    d[16, 16] = 1.0
    raw = ic.img_psf_conv(img, d)
    assert rel_err(raw, g["raw_delta"]) < 1e-6
    idx = np.maximum(np.arange(32) - 1, 0)
    assert np.abs(g["raw_delta"] - g["img"][:, :, idx][:, :, :, idx]).max() < 1e-5


def test_ic_real_size(volume_896):
    g = load_golden("ic_real.npz")
    vol = torch.from_numpy(np.ascontiguousarray(volume_896))
    assert rel_err(vol[:, ::16, ::16], g["volume_sub"]) == 0.0
    torch.manual_seed(int(g["noise_seed"]))
    noise = torch.rand([1, 896, 896, 1])
    m1, m2 = ic.disk_masks()
    img = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    w = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    for tag in ("init", "modelpth"):
        coeffs = torch.tensor(g[f"{tag}_coeffs"]).reshape(-1, 1, 1).requires_grad_(True)
        sensor, psf, loss = ic.forward(img, coeffs, vol, noise, prueba="3", mask_1=m1, mask_2=m2,
                                       height_tolerance=2e-8, sensor_distance=0.025, sample_interval=3e-6)
        assert psf.dtype == torch.float64 and loss.dtype == torch.float64
        assert rel_err(psf.detach(), g[f"{tag}_psf"]) < 1e-4
        assert abs(loss.item() - float(g[f"{tag}_loss"])) < 1e-5 * float(g[f"{tag}_loss"])
        assert rel_err(sensor.detach()[:, :, ::8, ::8], g[f"{tag}_sensor_sub"]) < 1e-4
        assert rel_err(sensor.detach()[:, :, :3, :], g[f"{tag}_sensor_edge"]) < 1e-4
        gs, = torch.autograd.grad((sensor * w).sum(), coeffs, retain_graph=True)
        gl, = torch.autograd.grad(loss, coeffs)
        assert rel_err(gs.reshape(-1)[3:], g[f"{tag}_grad_sensor_w"]) < TOL
        assert rel_err(gl.reshape(-1)[3:], g[f"{tag}_grad_loss"]) < TOL


def test_fd_camera():
    g = load_golden("fd.npz")
    for n, terms in ((64, 21), (256, 300)):
        t = f"n{n}"
        c = fd.constants(n)
        vol = torch.tensor(oz.zernike_volume(n, terms), dtype=torch.float32)
        if n == 64:
            assert rel_err(vol, g["n64_volume"]) == 0.0
        zt = torch.tensor(g[f"{t}_zer_train"]).requires_grad_(True)
        coeffs = torch.cat([torch.zeros(3, 1, 1), zt], 0)
        img = torch.rand(2, 3, n, n, generator=torch.Generator().manual_seed(0)) * 2 - 1
        w = torch.rand(2, 3, n, n, generator=torch.Generator().manual_seed(5))
        sensor, psfs, loss_rad, cl = fd.forward(c, img, coeffs, vol)
        assert rel_err(psfs.detach(), g[f"{t}_psfs"]) < 1e-5
        assert abs(loss_rad.item() - float(g[f"{t}_loss_rad"])) < 1e-5 * float(g[f"{t}_loss_rad"])
        assert abs(cl.item() - float(g[f"{t}_centering_loss"])) < 1e-4 * float(g[f"{t}_centering_loss"])
        s = max(1, n // 32)
        assert rel_err(sensor.detach()[:, :, ::s, ::s], g[f"{t}_sensor_sub"]) < 1e-4
        grad, = torch.autograd.grad((sensor * w).sum() + 1e3 * loss_rad + 1e6 * cl, zt)
        assert rel_err(grad.reshape(-1), g[f"{t}_grad"]) < TOL
