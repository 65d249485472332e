"""Drop-in for reference ``Face-DeId/Camera/Optics.py:9 Camera`` on MI355X (forward path).

Same constructor, parameters (``Zer_no_train``, ``Zer_train``, ``ca``), ``forward(img) -> img_sensor`` and the
side-channel attributes ``psfs``, ``loss_rad``, ``centering_loss`` (Optics.py:73-77,113,124-125).  The
input-independent chirps / apertures of ``get_psf`` are built once on the host with the reference's own float32
formulas (Optics.py:13-55,94-107) and kept resident; per call the height map, the 3-D FFT Fresnel step, the PSF,
both losses, the circular FFT convolution and the per-image amax normalisation run in libppv_hip.so.

Autograd: ``img_sensor``, ``loss_rad`` and ``centering_loss`` carry a graph to ``Zer_train`` (hand-written adjoint
kernels: sensor -> PSF correlation, PSF/losses -> height map through the adjoint 3-D FFT chain, height map -> Zernike
coefficients).  The gradient w.r.t. the input image is not provided (the reference's caller detaches, ``solver.py:144``).
"""
import numpy as np
import torch
from torch import nn

from . import _lib
from . import fftconv as fc
from ._lib import check, ptr, stream_ptr
from .zernike import zernike_volume


def _deta(lb):
    lens = torch.sqrt(1 + (0.6961663 * (lb ** 2) / ((lb ** 2) - 0.0684043 ** 2)
                           + 0.4079426 * (lb ** 2) / ((lb ** 2) - 0.1162414 ** 2)
                           + 0.8974794 * (lb ** 2) / ((lb ** 2) - 9.896161 ** 2)))
    air = 1 + 0.05792105 / (238.0185 - lb ** -2) + 0.00167917 / (57.362 - lb ** -2)
    return torch.abs(lens - air)


def _cexp(ph):
    return torch.complex(torch.cos(ph), torch.sin(ph))


class _FdPsfFn(torch.autograd.Function):
    """coeffs [K,1,1] -> (psf [1,3,N,N] f32, loss_rad f32 0-d, centering_loss f32 0-d)   (Optics.py:92-120,124-125)"""

    @staticmethod
    def forward(ctx, coeffs, cam):
        L = _lib.lib()
        N = cam.N
        c = coeffs.detach().reshape(-1).contiguous()
        h = torch.empty((N, N), dtype=torch.float32, device=cam.device)
        check(L.ppv_zernike_contract(ptr(cam.zernike_volume), ptr(c), ptr(h), c.numel(), h.numel(), stream_ptr()), "ppv_zernike_contract")
        psf = torch.empty((1, 3, N, N), dtype=torch.float32, device=cam.device)
        acc = torch.empty(4, dtype=torch.float64, device=cam.device)
        cam._state_token += 1
        check(L.ppv_fd_psf_fwd(ptr(h), ptr(cam._base), ptr(cam._chirp1), ptr(cam._chirp2T), ptr(cam._chirp3), ptr(cam._rho),
                               cam._kf_p, cam._lratio, cam._amp, ptr(psf), ptr(acc), ptr(cam._ws), N, stream_ptr()), "ppv_fd_psf_fwd")
        ctx.cam, ctx.token = cam, cam._state_token
        ctx.save_for_backward(psf, h, acc)
        ctx.set_materialize_grads(False)
        return psf, torch.sqrt(acc[0]).to(torch.float32), ((acc[1] + acc[2]) / (3.0 * N * N)).to(torch.float32)

    @staticmethod
    def backward(ctx, g_psf, g_lr, g_cl):
        cam = ctx.cam
        if ctx.token != cam._state_token:
            raise RuntimeError("Camera: forward() ran again before backward(); the saved optical state was overwritten")
        psf, h, acc = ctx.saved_tensors
        L = _lib.lib()
        N, K = cam.N, cam.zernike_volume.shape[0]
        dev = cam.device
        gp = g_psf.contiguous().float() if g_psf is not None else None
        glr = g_lr.to(torch.float64).contiguous() if g_lr is not None else None
        gcl = g_cl.to(torch.float64).contiguous() if g_cl is not None else None
        gh = torch.empty(N * N, dtype=torch.float32, device=dev)
        ws2 = torch.empty(L.ppv_fd_psf_bwd_workspace_bytes(N), dtype=torch.uint8, device=dev)
        check(L.ppv_fd_psf_bwd(ptr(gp), ptr(glr), ptr(gcl), ptr(psf), ptr(cam._base), ptr(cam._chirp1), ptr(cam._chirp2T),
                               ptr(cam._chirp3), ptr(cam._rho), cam._kf_p, cam._lratio, cam._amp, ptr(h), ptr(acc), ptr(gh),
                               ptr(cam._ws), ptr(ws2), N, stream_ptr()), "ppv_fd_psf_bwd")
        gc = torch.empty(K, dtype=torch.float32, device=dev)
        part = torch.empty(L.ppv_zernike_grad_scratch_bytes(K, N * N), dtype=torch.uint8, device=dev)
        check(L.ppv_zernike_grad(ptr(cam.zernike_volume), ptr(gh), ptr(gc), ptr(part), K, N * N, stream_ptr()), "ppv_zernike_grad")
        return gc.reshape(K, 1, 1), None


class _FdSensorFn(torch.autograd.Function):
    """(img [B,3,N,N], psf [1,3,N,N]) -> img (*) roll(psf) / per-image amax   (Optics.py:126-128)"""

    @staticmethod
    def forward(ctx, img, psf, cam):
        N = cam.N
        img = img.contiguous()
        otf = fc.otf_build(psf.detach()[0], N, N)
        out, _, partial = fc.fftconv_fwd(img, otf, mode=1)
        m = fc.group_max(partial, img.shape[0])
        fc.div_by_group_(out, m)
        ctx.save_for_backward(img, out, m)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("gradient w.r.t. the input image of the FD camera (the reference detaches, solver.py:144)")
        img, sensor, m = ctx.saved_tensors
        L = _lib.lib()
        B, C, N, _ = img.shape
        g_psf = torch.empty((1, C, N, N), dtype=torch.float32, device=img.device)
        ws = torch.empty(L.ppv_fftconv_fd_bwd_workspace_bytes(B, C, N), dtype=torch.uint8, device=img.device)
        check(L.ppv_fftconv_fd_bwd(ptr(img), ptr(g.contiguous()), ptr(sensor), ptr(m), ptr(g_psf), ptr(ws), B, C, N, stream_ptr()),
              "ppv_fftconv_fd_bwd")
        return None, g_psf, None


class Camera(nn.Module):
    supports_backward = True

    def __init__(self, device="cpu", N=256, lamdas=3, zernike_terms=50, height_tolerance=2e-8):
        super().__init__()
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("ppv_amd Camera runs on an MI355X (device must be cuda); no CPU path")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if N not in (256, 512):
            raise NotImplementedError("N in {256, 512} are compiled in")
        self.device, self.N, self.c, self.lamdas, self.height_tolerance = device, N, N // 2, lamdas, height_tolerance
        # ---- Optics.py:13-55 on the host, float32 as the reference
        self.zi, self.z0 = 50e-3, 5.
        self.f = 1 / (1 / self.zi + 1 / self.z0)
        self.R = self.f * _deta(torch.tensor(550e-9 * 1e6))
        self.radii = 2.0e-3
        pi = torch.tensor([np.pi])
        self.L_len = 2 * self.radii * 2
        self.px = 3.713103e-6
        self.L_sen = self.px * N
        lamb = (torch.tensor([640, 550, 440]) * 1.e-9).unsqueeze(-1).unsqueeze(-1)
        flmb = self.R / _deta(lamb * 1e6)
        k = 2 * pi / lamb
        self.z = torch.tensor([0.75])
        du = self.L_len / N
        u = torch.arange(-1 * self.L_len / 2, self.L_len / 2, du)
        X, Y = torch.meshgrid(u, u, indexing="ij")
        XY = X * X + Y * Y
        rad = torch.sqrt(X ** 2 + Y ** 2) <= self.radii
        fx1 = torch.arange(-1 / (2 * du), 1 / (2 * du), 1 / self.L_len)
        fx1 = torch.roll(fx1, -(fx1.numel() // 2), 0)
        FX, FY = torch.meshgrid(fx1, fx1, indexing="ij")
        FF = FX * FX + FY * FY
        dx2 = self.L_sen / N
        x2 = torch.arange(-1 * self.L_sen / 2, self.L_sen / 2, dx2)
        X2, Y2 = torch.meshgrid(x2, x2, indexing="ij")
        XY2 = X2 * X2 + Y2 * Y2
        rho = (torch.sqrt(X2 ** 2 + Y2 ** 2) > self.px * 32) * 1.
        if XY.shape != (N, N) or FF.shape != (N, N) or XY2.shape != (N, N):
            raise RuntimeError("sampling grids do not have N points (float arange length)")
        dis = self.z[0]
        t = _cexp(-(k / (2 * flmb)) * XY)                                              # Optics.py:95
        focus = _cexp((k / (2 * dis)) * XY)                                            # :96
        base = torch.mul(rad, torch.mul(t, focus))                                     # :98 (left factor)
        chirp1 = _cexp((pi / (lamb * self.zi * self.L_len) * (self.L_len - self.L_sen)) * XY)      # :100
        chirp2 = _cexp(-(pi * lamb * self.zi * self.L_len / self.L_sen) * FF)          # :103
        chirp3 = _cexp(-(pi / (lamb * self.zi * self.L_sen) * (self.L_len - self.L_sen)) * XY2)    # :107
        self._base = base.to(torch.complex64).contiguous().to(device)
        self._chirp1 = chirp1.contiguous().to(device)
        self._chirp2T = chirp2.permute(0, 2, 1).contiguous().to(device)
        self._chirp3 = chirp3.contiguous().to(device)
        self._rho = rho.to(torch.float32).contiguous().to(device)
        self._kf = np.ascontiguousarray((k * flmb).reshape(-1).numpy().astype(np.float32))
        self._kf_p = self._kf.ctypes.data_as(_lib.ctypes.c_void_p)
        self._lratio = float(self.L_sen / self.L_len)
        self._amp = float((du * du) / (dx2 * dx2))
        self.k, self.flmb, self.lamb = k.to(device), flmb.to(device), lamb.to(device)
        # ---- parameters (Optics.py:59-70): same RNG draws in the same order
        zernike_inits = torch.rand((zernike_terms, 1, 1), device=device) / 100
        zernike_inits[:3] = 0
        self.Zer_no_train = nn.Parameter(zernike_inits[:3, ...], requires_grad=False)
        self.Zer_train = nn.Parameter(zernike_inits[3:, ...], requires_grad=True)
        self.zernike_volume = zernike_volume(N, zernike_terms, device)
        size = (1, 1, 32, 32)
        self.ca = nn.Parameter(torch.where(torch.rand(size=size) > 0.5, torch.ones(size), torch.zeros(size)).to(device),
                               requires_grad=False)
        self.loss_psf, self.loss_rad, self.psfs, self.centering_loss, self.psf_rad = 0.0, 0.0, None, None, None
        L = _lib.lib()
        with torch.cuda.device(device):
            check(L.ppv_init(), "ppv_init")
        self._ws = torch.empty(L.ppv_fd_psf_workspace_bytes(N), dtype=torch.uint8, device=device)
        self._state_token = 0

    def __getstate__(self):                        # whole-module pickling (solver checkpoints / DataParallel replicas)
        st = dict(self.__dict__)
        st.pop("_kf_p", None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._kf_p = self._kf.ctypes.data_as(_lib.ctypes.c_void_p)

    def get_Heith_Map(self):
        c = torch.cat((self.Zer_no_train, self.Zer_train), 0).detach().reshape(-1).contiguous()
        h = torch.empty((self.N, self.N), dtype=torch.float32, device=self.device)
        check(_lib.lib().ppv_zernike_contract(ptr(self.zernike_volume), ptr(c), ptr(h), c.numel(), h.numel(), stream_ptr()),
              "ppv_zernike_contract")
        return h.unsqueeze(0)

    def get_phase_shift(self):
        return self.k * self.flmb * self.get_Heith_Map()

    def load_ckpt(self):
        ckpt = torch.load('./Camera/Cam_focus.pth', map_location=self.device)
        self.load_state_dict(ckpt['camera'])

    def get_psf(self):
        coeffs = torch.cat((self.Zer_no_train, self.Zer_train), 0)
        self.psfs, self.loss_rad, self._centering = _FdPsfFn.apply(coeffs, self)
        return self.psfs

    def forward(self, img):
        if img.device != self.device:
            raise RuntimeError("input must live on the module's MI355X device")
        if img.shape[-1] != self.N or img.shape[-2] != self.N:
            raise RuntimeError("image size must equal the camera's N (Optics.py:124-126 rolls by img.size // 2)")
        psf = self.get_psf()
        self.centering_loss = self._centering                                          # Optics.py:124-125
        return _FdSensorFn.apply(img.to(torch.float32), psf, self)                     # Optics.py:126-128
