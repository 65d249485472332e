"""GPU parity: HIP FFT image(x)PSF convolution vs the CPU oracle (same seeded inputs), through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _psf(P, seed):
    g = torch.Generator().manual_seed(seed)
    psf = torch.rand(1, P, P, 3, generator=g, dtype=torch.float64) ** 4
    return psf / psf.sum(dim=[1, 2], keepdim=True)


@pytest.mark.parametrize("P,B", [(256, 3), (128, 2)])
def test_ic_img_psf_conv(P, B):
    import ppv_amd.fftconv as fc
    from oracle import ic_camera as ic
    img = torch.rand(B, 3, P, P, generator=torch.Generator().manual_seed(0))
    psf = _psf(P, 1)
    want = ic.img_psf_conv(img, psf.permute(1, 2, 0, 3).to(torch.float32))
    psf_d = psf.cuda()
    otf = fc.otf_build(psf_d[0].permute(2, 0, 1), P, 2 * P)
    got, signs, partial = fc.fftconv_fwd(img.cuda(), otf, mode=0)
    torch.cuda.synchronize()
    assert rel_err(got.cpu(), want) < 1e-5
    assert abs(partial.max().item() - want.max().item()) < 1e-5 * want.max().item()
    m = fc.group_max(partial, 1)
    fc.div_by_group_(got, m)
    assert rel_err(got.cpu(), want / want.max()) < 1e-5
    assert (signs.cpu() == 0).all()          # non-negative image and PSF


@pytest.mark.parametrize("P,B", [(368, 2), (300, 1), (130, 2), (64, 3), (512, 1)])
def test_ic_img_psf_conv_at_any_even_patch_size(P, B):
    """Patch sizes off the 128 / 256 grid (the reference constructor's default is 368, Lens.py:21-22) on the next 256 / 512 / 1024-point
    transform (ppv_fftconv_ic_fwd_p) against the oracle's 2 P-point convolution (Utils.py:251-297), a signed PSF included (sign bits)."""
    import ppv_amd.fftconv as fc
    from oracle import ic_camera as ic
    img = torch.rand(B, 3, P, P, generator=torch.Generator().manual_seed(0))
    psf = _psf(P, 1) - (0.3 / (P * P) if P == 300 else 0.0)
    want = ic.img_psf_conv(img, psf.permute(1, 2, 0, 3).to(torch.float32))
    N = fc.ic_transform_length(P)
    assert N >= 2 * P and N in (256, 512, 1024)
    otf = fc.otf_build(psf.cuda()[0].permute(2, 0, 1), P, N)
    got, signs, partial = fc.fftconv_ic_fwd(img.cuda(), otf, N)
    assert rel_err(got.cpu(), want) < 2e-5
    assert abs(partial.max().item() - want.max().item()) < 2e-5 * want.max().item()
    # delta PSF: the reference's integer index path out(i, j) = img(max(i-1, 0), max(j-1, 0))
    d = torch.zeros(3, P, P, device="cuda")
    d[:, P // 2, P // 2] = 1.0
    got_d, _, _ = fc.fftconv_ic_fwd(img.cuda(), fc.otf_build(d, P, N), N)
    idx = np.maximum(np.arange(P) - 1, 0)
    assert np.abs(got_d.cpu().numpy() - img.numpy()[:, :, idx][:, :, :, idx]).max() < 4e-6


def test_ic_delta_psf_index_map():
    """delta PSF at the centre: out(i,j) == img(max(i-1,0), max(j-1,0)) -- the reference's integer index path."""
    import ppv_amd.fftconv as fc
    P = 256
    img = torch.rand(2, 3, P, P, generator=torch.Generator().manual_seed(3))
    psf = torch.zeros(3, P, P, device="cuda")
    psf[:, P // 2, P // 2] = 1.0
    otf = fc.otf_build(psf, P, 2 * P)
    got, _, _ = fc.fftconv_fwd(img.cuda(), otf, mode=0)
    idx = np.maximum(np.arange(P) - 1, 0)
    want = img.numpy()[:, :, idx][:, :, :, idx]
    assert np.abs(got.cpu().numpy() - want).max() < 2e-6


@pytest.mark.parametrize("N,B", [(256, 2), (512, 2)])
def test_fd_conv2d(N, B):
    import ppv_amd.fftconv as fc
    from oracle import fd_camera as fd
    img = torch.rand(B, 3, N, N, generator=torch.Generator().manual_seed(0)) * 2 - 1
    psf = _psf(N, 2)[0].permute(2, 0, 1).to(torch.float32).contiguous()     # [3,N,N], centre at N/2
    rolled = torch.roll(psf, shifts=(-(N // 2), -(N // 2)), dims=(-2, -1))
    want = fd.conv2d_circular(img, rolled[None])
    otf = fc.otf_build(psf.cuda(), N, N)
    got, _, partial = fc.fftconv_fwd(img.cuda(), otf, mode=1)
    assert rel_err(got.cpu(), want) < 1e-5
    m = fc.group_max(partial, B)
    assert rel_err(m.cpu(), want.amax((1, 2, 3))) < 1e-5
    fc.div_by_group_(got, m)
    assert rel_err(got.cpu(), want / want.amax((1, 2, 3))[:, None, None, None]) < 1e-5


def test_golden_tiny_otf_geometry():
    """The reference's own delta-PSF output (golden, P=32) has the same index map our kernels implement."""
    g = load_golden("ic_tiny.npz")
    idx = np.maximum(np.arange(32) - 1, 0)
    assert np.abs(g["raw_delta"] - g["img"][:, :, idx][:, :, :, idx]).max() < 1e-5
