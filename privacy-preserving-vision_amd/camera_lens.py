"""Drop-in for reference ``Image_Caption/Camera/Lens.py:11 OpticsZernike`` on MI355X.

Same constructor signature, ``forward()`` signature and 4-tuple return, parameter names and
``state_dict`` keys; the compute runs in libppv_hip.so (csrc/psf_ic.hip, csrc/fftconv.hip).
Input-independent constants (spherical wavefront Lens.py:191-210, Fresnel transfer function
Utils.py:339-373, disk masks Lens.py:111-127) are built ONCE here on the host with the
reference's own fp64 formulas and kept resident; the reference rebuilds them every call.
"""
import os

import numpy as np
import torch
from torch import nn

from . import _lib
from . import fftconv as fc
from ._lib import check, ptr, stream_ptr
from .zernike import zernike_volume


def _c64_of_phase(phase64):
    return torch.complex(torch.cos(phase64).to(torch.float32), torch.sin(phase64).to(torch.float32))


def _disk(size, radius):
    yy, xx = np.mgrid[0:size, 0:size]
    return ((xx - size // 2) ** 2 + (yy - size // 2) ** 2) <= radius * radius


class _IcPsfFn(torch.autograd.Function):
    """coeffs [K,1,1] f32 -> (psf_n [1,P,P,3] f32, psf_m [1,P,P,3] f64 | empty, loss f64 0-d | empty)."""

    @staticmethod
    def forward(ctx, coeffs, noise, cam, use_m1, use_m2):
        L = _lib.lib()
        dev = coeffs.device
        P, RR, K = cam.patch_size, cam.wave_res[0], cam.zernike_volume.shape[0]
        psf_n = torch.empty((1, P, P, 3), dtype=torch.float32, device=dev)
        psf_m = torch.empty((1, P, P, 3), dtype=torch.float64, device=dev) if use_m2 else None
        loss_acc = torch.zeros((), dtype=torch.float64, device=dev) if use_m1 else None
        c = coeffs.detach().reshape(-1).contiguous()
        tol = float(cam.height_tolerance) if cam.height_tolerance is not None else -1.0
        cam._state_token += 1
        check(L.ppv_ic_psf_fwd(ptr(cam.zernike_volume), ptr(c), ptr(noise), ptr(cam._sph), ptr(cam._Ht), cam._kdn_p,
                               tol, ptr(cam.mask_1 if use_m1 else None), ptr(cam.mask_2 if use_m2 else None),
                               ptr(psf_n), ptr(psf_m), ptr(loss_acc), ptr(cam._state), RR, P, K, cam._up,
                               cam._up_scale, stream_ptr()), "ppv_ic_psf_fwd")
        loss = torch.sqrt(loss_acc) if use_m1 else None
        ctx.cam, ctx.token, ctx.use_m1, ctx.use_m2 = cam, cam._state_token, use_m1, use_m2
        ctx.save_for_backward(psf_n, loss)
        ctx.set_materialize_grads(False)
        empty = torch.empty(0, device=dev)
        return psf_n, (psf_m if use_m2 else empty), (loss if use_m1 else empty)

    @staticmethod
    def backward(ctx, g_psf_n, g_psf_m, g_loss):
        cam = ctx.cam
        if ctx.token != cam._state_token:
            raise RuntimeError("OpticsZernike: forward() ran again before backward(); the saved optical state "
                               "(one per module) was overwritten")
        psf_n, loss = ctx.saved_tensors
        L = _lib.lib()
        P, RR, K = cam.patch_size, cam.wave_res[0], cam.zernike_volume.shape[0]
        g_c = torch.empty(K, dtype=torch.float32, device=psf_n.device)
        gm = g_psf_m.contiguous() if (ctx.use_m2 and g_psf_m is not None) else None
        gn = g_psf_n.contiguous() if g_psf_n is not None else None
        gl = g_loss.to(torch.float64).contiguous() if (ctx.use_m1 and g_loss is not None) else None
        check(L.ppv_ic_psf_bwd(ptr(cam.zernike_volume), ptr(cam._Ht), cam._kdn_p,
                               ptr(cam.mask_1 if ctx.use_m1 else None), ptr(cam.mask_2 if ctx.use_m2 else None),
                               ptr(psf_n), ptr(gm), ptr(gn), ptr(gl), ptr(loss), ptr(g_c), ptr(cam._state),
                               RR, P, K, cam._up, cam._up_scale, stream_ptr()), "ppv_ic_psf_bwd")
        return g_c.reshape(K, 1, 1), None, None, None, None


class _IcSensorFn(torch.autograd.Function):
    """(img [B,3,P,P] f32, psf [1,P,P,3] f32|f64) -> sensor = |img (x) psf| / max   (Lens.py:290,312)."""

    @staticmethod
    def forward(ctx, img, psf, cam):
        B, C, P, _ = img.shape
        N = fc.ic_transform_length(P)          # 2 P for the 128 / 256 patches, the next of 256 / 512 / 1024 otherwise (same convolution)
        img = img.contiguous()
        otf = fc.otf_build(psf.detach()[0].permute(2, 0, 1), P, N)
        out, signs, partial, ws = fc.fftconv_ic_fwd(img, otf, N, return_workspace=True)
        # the forward workspace starts with the row transform of the image: kept (2 x 0.2 GB at B = 128) when the PSF needs a gradient,
        # so that backward does not recompute it (PPV_IC_KEEP_ROWS=0: recompute)
        ctx.rows_ws = ws if (ctx.needs_input_grad[1] and os.environ.get("PPV_IC_KEEP_ROWS", "1") != "0") else None
        m = fc.group_max(partial, 1)
        if cam.global_max_sync and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MAX)
        fc.div_by_group_(out, m)
        ctx.save_for_backward(img, out, signs, m, otf)
        ctx.cam, ctx.psf_meta = cam, (psf.dtype, psf.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        img, sensor, signs, m, otf = ctx.saved_tensors
        L = _lib.lib()
        B, C, P, _ = img.shape
        N = fc.ic_transform_length(P)
        dev = img.device
        g = g.contiguous()
        dotcnt = torch.empty(2, dtype=torch.float64, device=dev)
        check(L.ppv_sensor_dot_count(ptr(g), ptr(sensor), ptr(dotcnt), g.numel(), stream_ptr()), "ppv_sensor_dot_count")
        if ctx.cam.global_max_sync and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(dotcnt)
        need_img, need_psf = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dtype, shape = ctx.psf_meta
        g_psf = torch.empty(shape, dtype=dtype, device=dev) if need_psf else None
        g_img = torch.empty_like(img) if need_img else None
        ws = torch.empty(L.ppv_fftconv_ic_bwd_workspace_bytes_p(B, C, P, N), dtype=torch.uint8, device=dev)
        sc, sy, sx = (1, P * 3, 3)       # [1,P,P,3] viewed as [C][P][P]
        check(L.ppv_fftconv_ic_bwd_p(ptr(img), int(img.dtype == torch.uint8), ptr(g), ptr(sensor), ptr(signs), ptr(m), ptr(dotcnt), ptr(otf),
                                     ptr(g_psf), int(dtype == torch.float64), sc, sy, sx, ptr(g_img), ptr(ws),       # uint8 pixels: no image gradient
                                     ptr(ctx.rows_ws), B, C, P, N, stream_ptr()), "ppv_fftconv_ic_bwd_p")
        ctx.rows_ws = None
        return g_img, g_psf, None


class OpticsZernike(nn.Module):
    """See module docstring.  Extra keyword-only arguments (not in the reference):

    coeff_layout     "A" (committed Lens.py:92-96: one trainable defocus coefficient + ``zernike_coeffs_no_train2``)
                     or "B" (the layout of the committed ``Camera/Model.pth`` and of the commented-out code
                     Lens.py:99-101: ``zernike_coeffs_train`` is (K-3,1,1)).  ``load_state_dict`` switches
                     automatically to the layout of the checkpoint it is given.
    zernike_volume   optional precomputed [K,R,R] float32 tensor (else ./zernike_volumes/*.npy cache as the
                     reference, else generated on the GPU).
    global_max_sync  all-reduce(MAX) the normalising maximum of Lens.py:312 across data-parallel ranks.
    """

    def __init__(self, input_shape, device, experiment=None, sensor_distance=25e-3,
                 refractive_idcs=np.array([1.499, 1.493, 1.488]), wave_lengths=np.array([460, 550, 640]) * 1e-9,
                 height_tolerance=20e-9, wave_resolution=(736, 736), patch_size=368, sample_interval=2e-6,
                 upsample=False, frames=8, optics_cfg=1, zernike_terms=350, mask_1=None, mask_2=None, *,
                 coeff_layout="A", zernike_volume_tensor=None, global_max_sync=False):
        super().__init__()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ppv_amd OpticsZernike runs on an MI355X (device must be cuda); no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.sensor_distance = sensor_distance
        self.height_tolerance = height_tolerance
        self.upsample = upsample
        self.patch_size = patch_size
        self.wave_lengths = np.asarray(wave_lengths, dtype=np.float64)
        self.sample_interval = sample_interval
        self.refractive_idcs = np.asarray(refractive_idcs, dtype=np.float64)
        self.frames = frames
        self.optics_cfg = optics_cfg
        self.zernike_terms = zernike_terms
        self.global_max_sync = global_max_sync
        self.wave_res = [patch_size * 4, patch_size * 4] if wave_resolution is None else list(wave_resolution)
        self.physical_size = float(self.wave_res[0] * self.sample_interval)
        self.channels = input_shape[-1]
        if upsample:
            raise NotImplementedError("upsample=True (Lens.py:286-293) is outside the MI355X hot path")
        if len(self.wave_lengths) != 3:
            raise NotImplementedError("three wavelengths (RGB) are compiled in")
        RR, P = self.wave_res[0], patch_size
        if self.wave_res[0] != self.wave_res[1] or P % 2 or RR % 4:
            raise NotImplementedError("square wave resolution (a multiple of 4) and an even patch_size are compiled in")
        # The image convolution kernels (csrc/fftconv.hip) run 256-, 512- and 1024-point transforms.  patch_size 128 / 256 (the reference's
        # scripts pass 256, train.py:64-66) fill a 2 P-point transform exactly; any other even patch size <= 512 -- the constructor's own
        # default 368 (Lens.py:22) -- runs on the next of those lengths: image and PSF have support P x P, so every transform of at least
        # 2 P - 1 points computes the reference's 2 P-point circular convolution (round 5: this used to go through a library FFT).  The PSF
        # (Fresnel propagation on the 1.5 * RR grid) is native for every RR whose padded length factors into 2, 3, 5, 7, 11, 13, 23.
        if P > 512:
            raise NotImplementedError("patch_size above 512: the image convolution kernels stop at 1024-point transforms")

        # --- Zernike volume (Lens.py:66-78): same on-disk cache name as the reference, else GPU generator
        cache = 'zernike_volumes/zernike_volume_%d_n%d.npy' % (RR, zernike_terms)
        if zernike_volume_tensor is not None:
            vol = zernike_volume_tensor.to(self.device, torch.float32)
        elif os.path.exists(cache):
            vol = torch.tensor(np.load(cache), dtype=torch.float32, device=self.device)
        else:
            vol = zernike_volume(RR, zernike_terms, self.device)
        self.zernike_volume = vol.contiguous()
        K = self.zernike_volume.shape[0]

        inits = np.zeros((K, 1, 1))
        inits[3] = -22                                                        # Lens.py:88-90
        self.coeff_layout = coeff_layout
        self._make_params(torch.tensor(inits, dtype=torch.float32))

        self.training_info = None
        self.gpu_rank = None
        self.experiment = experiment

        # --- disk masks (Lens.py:111-127): hard-wired 256 x 256 x 3 float64, radius 32 (Euclidean disk; the
        #     reference rasterises with cv2.circle -- boundary pixels unpinned, masks stay overridable attributes)
        d = _disk(256, 32)
        m1 = np.repeat((~d).astype(np.float64)[:, :, None], 3, axis=2)
        m2 = np.repeat(d.astype(np.float64)[:, :, None], 3, axis=2)
        self.mask_1 = torch.from_numpy(m1).to(self.device)
        self.mask_2 = torch.from_numpy(m2).to(self.device)

        # --- cached optics constants, built with the reference's fp64 host formulas
        n_, m_ = self.wave_res
        x, y = np.mgrid[-n_ // 2:n_ // 2, -m_ // 2:m_ // 2].astype(np.float64)
        x = x / n_ * self.physical_size
        y = y / m_ * self.physical_size
        depth = 1 / 2 if optics_cfg == 1 else 1                               # Lens.py:202-205
        wave_nos = torch.tensor((2.0 * np.pi / self.wave_lengths).reshape([1, 1, 1, -1]))
        curv = torch.sqrt(torch.tensor(x ** 2 + y ** 2) + torch.tensor(depth, dtype=torch.float64) ** 2)
        sph = _c64_of_phase(wave_nos * curv.unsqueeze(0).unsqueeze(-1))      # [1,RR,RR,3] c64
        self._sph = sph[0].contiguous().to(self.device)
        pad = RR // 4
        M = RR + 2 * pad
        fx_i, fy_i = np.mgrid[-M // 2:M // 2, -M // 2:M // 2]
        fx = np.fft.ifftshift(fx_i / (self.sample_interval * M))
        fy = np.fft.ifftshift(fy_i / (self.sample_interval * M))
        sq = (np.square(fx) + np.square(fy))[None, :, :, None]
        if torch.is_tensor(sensor_distance):
            raise NotImplementedError("trainable sensor_distance (Utils.py:356-362)")
        expo = np.float64(self.wave_lengths * np.pi * -1.0 * sq * self.sensor_distance)
        H = _c64_of_phase(torch.tensor(expo, dtype=torch.float64))           # [1,M,M,3]
        self._Ht = H[0].permute(2, 1, 0).contiguous().to(self.device)        # [3][kx][ky]
        kdn = (2.0 * np.pi / self.wave_lengths) * (self.refractive_idcs - 1.0)
        self._kdn = np.ascontiguousarray(kdn, dtype=np.float64)
        self._kdn_p = self._kdn.ctypes.data_as(_lib.ctypes.c_void_p)
        # area down-sampling geometry (Utils.py:220-246)
        if RR % P == 0:
            self._up, self._up_scale = RR // P, 1.0
        else:
            lcm = abs(P * RR) / np.gcd(P, RR) / P
            self._up = 10 if lcm > 10 else int(lcm)
            self._up_scale = float(np.float32(RR) / np.float32(self._up * P))
        L = _lib.lib()
        with torch.cuda.device(self.device):
            check(L.ppv_init(), "ppv_init")
        self._state = torch.empty(L.ppv_ic_psf_state_bytes(RR, P, K), dtype=torch.uint8, device=self.device)
        self._state_token = 0
        # the basis is zero outside the aperture disk (poppy zernike_basis(outside=0), Utils.py:75-77): mark its support once so the two
        # 1.12-GB passes over it skip those pixels (exact; derived from the DATA, so a user-supplied volume is handled too)
        with torch.cuda.device(self.device):
            check(L.ppv_ic_psf_state_init(ptr(self._state), RR, P, K, stream_ptr()), "ppv_ic_psf_state_init")
            check(L.ppv_ic_psf_mark_support(ptr(self.zernike_volume), ptr(self._state), RR, P, K, stream_ptr()), "ppv_ic_psf_mark_support")

    # ------------------------------------------------------------------ parameters / checkpoints
    def _make_params(self, full):
        dev = self.device
        self.zernike_coeffs_no_train = nn.Parameter(full[:3].clone().to(dev), requires_grad=False)
        if self.coeff_layout == "A":
            self.zernike_coeffs_no_train2 = nn.Parameter(full[4:].clone().to(dev), requires_grad=False)
            self.zernike_coeffs_train = nn.Parameter(full[3].clone().to(dev))
        elif self.coeff_layout == "B":
            if "zernike_coeffs_no_train2" in self._parameters:
                del self._parameters["zernike_coeffs_no_train2"]
            self.zernike_coeffs_no_train2 = None
            self.zernike_coeffs_train = nn.Parameter(full[3:].clone().to(dev))
        else:
            raise ValueError("coeff_layout must be 'A' or 'B'")

    def _concat(self):
        if self.coeff_layout == "A":                                          # Lens.py:158-159
            return torch.cat((self.zernike_coeffs_no_train, self.zernike_coeffs_train.unsqueeze(0),
                              self.zernike_coeffs_no_train2), 0)
        return torch.cat((self.zernike_coeffs_no_train, self.zernike_coeffs_train), 0)   # Lens.py:150,136

    def load_state_dict(self, state_dict, strict=True, **kw):
        tr = state_dict.get("zernike_coeffs_train")
        if tr is not None:
            want = "B" if tr.dim() == 3 else "A"
            if want != self.coeff_layout:
                # switch layout IN PLACE (``.data``) so an optimizer built before the load (train.py:66-78) keeps
                # pointing at the live parameter
                full = self._concat().detach().clone()
                self.coeff_layout = want
                if want == "B":
                    self._parameters.pop("zernike_coeffs_no_train2", None)
                    self.zernike_coeffs_no_train2 = None
                    self.zernike_coeffs_train.data = full[3:].clone()
                else:
                    self.__dict__.pop("zernike_coeffs_no_train2", None)
                    self.zernike_coeffs_no_train2 = nn.Parameter(full[4:].clone(), requires_grad=False)
                    self.zernike_coeffs_train.data = full[3].clone()
        return super().load_state_dict(state_dict, strict=strict, **kw)

    # whole-module pickling (the reference checkpoints the objects themselves, Image_Caption/utils.py:387-395): host pointers
    # into numpy constants are rebuilt on load
    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_kdn_p", None)
        return st

    def __setstate__(self, st):
        super().__setstate__(st)
        self._kdn_p = self._kdn.ctypes.data_as(_lib.ctypes.c_void_p)

    def get_Heith_Map(self):
        """Lens.py:129-139."""
        c = self._concat().detach().reshape(-1).contiguous()
        return torch.sum(c.reshape(-1, 1, 1) * self.zernike_volume, dim=0).unsqueeze(0)

    def load_pretrained_from_numpy(self, path):
        weights = np.load(path)['optics_trained_weights']
        self.state_dict()['zernike_coeffs_train'].copy_(torch.tensor(weights))

    def load_pretrained_from_warmup(self, path):
        ckpt = torch.load(os.path.expanduser(path), map_location=self.device)
        self.state_dict()['zernike_coeffs_train'].copy_(ckpt['model_state_dict']['optics.zernike_coeffs_train'])

    # ------------------------------------------------------------------ forward
    def forward(self, input_img, new_zernike=None, prueba=None, psf_lab=None, enfoco=None, *, noise_u01=None):
        if psf_lab is True:
            raise NotImplementedError("psf_lab=True reads a lab JPEG (Lens.py:222-233): out of scope")
        if input_img.device != self.device:
            raise RuntimeError("input must live on the module's MI355X device")
        coeffs = self._concat()
        if enfoco is True:                                                    # Lens.py:163-165
            coeffs = torch.zeros_like(coeffs)
            coeffs[3] = -22
        RR = self.wave_res[0]
        noise = None
        if self.height_tolerance is not None:                                 # Utils.py:403: drawn every call, train or eval
            noise = noise_u01 if noise_u01 is not None else torch.rand([1, RR, RR, 1], dtype=torch.float32,
                                                                       device=self.device)
            noise = noise.contiguous()
        use_m1 = prueba in ("1", "3")
        use_m2 = prueba in ("2", "3")
        if use_m1 or use_m2:
            # the kernels index the masks as [patch, patch, 3] through raw pointers: a hard-wired 256^2 mask (Lens.py:111-127) with
            # another patch size would be read with the wrong stride -- the reference fails on the broadcast there, so do we
            P = self.patch_size
            for m in ((self.mask_1,) if use_m1 else ()) + ((self.mask_2,) if use_m2 else ()):
                if tuple(m.shape) != (P, P, 3) or m.dtype != torch.float64 or m.device != self.device:
                    raise ValueError(f"prueba={prueba!r} needs mask_1 / mask_2 of shape ({P}, {P}, 3) float64 on {self.device}; got "
                                     f"{tuple(m.shape)} {m.dtype} (the reference's masks are hard-wired to 256 x 256, Lens.py:111-127)")
        psf_n, psf_m, loss = _IcPsfFn.apply(coeffs, noise, self, use_m1, use_m2)
        psf = psf_m if use_m2 else psf_n
        # uint8 input = the data set's raw pixels (HDF5 uint8, utils.py:94-150): decoded as x / 255 inside the first FFT kernel
        # (datasets.py:46 `imgs / 255.`), forward and backward; anything else is taken as float32 in [0, 1] like the reference
        sensor_img = _IcSensorFn.apply(input_img if input_img.dtype == torch.uint8 else input_img.to(torch.float32), psf, self)
        np.random.uniform(low=0.001, high=0.02)                               # Lens.py:295 (RNG stream parity; value unused)
        return sensor_img, psf, coeffs, (loss if use_m1 else None)


def conv2D(img, kernel):
    """Module-level helper of the reference (Lens.py:342-347): circular rfft2 convolution, stock torch.fft
    (not on the hot path; kept for import compatibility)."""
    return torch.fft.irfft2(torch.fft.rfft2(img, dim=(-2, -1)) * torch.fft.rfft2(kernel, dim=(-2, -1)), dim=(-2, -1))
