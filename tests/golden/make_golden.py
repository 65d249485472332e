#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by importing the REFERENCE's own Python.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py all

The reference needs three third-party modules that are not installed here and are not
part of /root/reference: ``torchvision`` (only
``transforms.Resize(interpolation=0)`` on tensors == legacy nearest), ``cv2`` (only
``circle`` filled) and ``poppy`` (only ``zernike.zernike_basis``).  Minimal stand-ins
for exactly those calls are registered in ``sys.modules`` below; the basis and the disk
come from ``oracle/zernike.py`` (parity unpinned vs poppy / OpenCV, see oracle/__init__.py).
Everything else that executes is the reference's code, unmodified, on torch CPU.
"""
import math
import zlib
import os
import subprocess
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def install_standins():
    from oracle import zernike as oz

    np.math = math                                    # numpy>=2 dropped np.math (IC Utils.py:213)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.models = types.ModuleType("torchvision.models")
    tv.utils = types.ModuleType("torchvision.utils")

    class Resize:
        def __init__(self, size, interpolation=0):
            assert interpolation == 0
            self.size = list(size)

        def __call__(self, x):
            return F.interpolate(x, size=self.size, mode="nearest")

    tv.transforms.Resize = Resize
    cv2 = types.ModuleType("cv2")
    cv2.FILLED = -1

    def circle(img, center, radius, color, thickness=-1, lineType=-1):
        d = oz.filled_disk(img.shape[0], center, radius)
        img[d] = color
        return img

    cv2.circle = circle
    poppy = types.ModuleType("poppy")
    poppy.zernike = types.ModuleType("poppy.zernike")
    poppy.zernike.zernike_basis = lambda nterms, npix, outside=0.0: oz.zernike_basis(nterms, npix, outside)
    for name, mod in [("torchvision", tv), ("torchvision.transforms", tv.transforms),
                      ("torchvision.models", tv.models), ("torchvision.utils", tv.utils),
                      ("cv2", cv2), ("poppy", poppy), ("poppy.zernike", poppy.zernike)]:
        sys.modules[name] = mod


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.max().item(), t.min().item()])


# --------------------------------------------------------------------------- IC
def gen_ic():
    install_standins()
    sys.path.insert(0, os.path.join(REF, "Image_Caption"))
    os.makedirs("/tmp/ppv_golden_scratch", exist_ok=True)
    os.chdir("/tmp/ppv_golden_scratch")               # the reference caches the basis as .npy in cwd
    from Camera.Lens import OpticsZernike
    from Camera import Utils as RU
    cpu = torch.device("cpu")

    def run(cam, img, prueba, seed_noise, w):
        for p in (cam.zernike_coeffs_train, cam.zernike_coeffs_no_train2):
            p.requires_grad_(True)
            p.grad = None
        torch.manual_seed(seed_noise)
        sensor, psf, coeffs, loss = cam(img, None, prueba)
        l_sensor = (sensor * w).sum()
        g_sensor = torch.autograd.grad(l_sensor, [cam.zernike_coeffs_train, cam.zernike_coeffs_no_train2],
                                       retain_graph=True)
        g_sensor = torch.cat([g_sensor[0].reshape(1), g_sensor[1].reshape(-1)])
        g_loss = None
        if loss is not None:
            g = torch.autograd.grad(loss, [cam.zernike_coeffs_train, cam.zernike_coeffs_no_train2])
            g_loss = torch.cat([g[0].reshape(1), g[1].reshape(-1)])
        return sensor, psf, coeffs, loss, g_sensor, g_loss

    # ---- G1: tiny, every stage -------------------------------------------------
    cam = OpticsZernike(input_shape=[None, 32, 32, 3], device=cpu, zernike_terms=15, patch_size=32,
                        height_tolerance=2e-8, sensor_distance=0.025, wave_resolution=[112, 112],
                        sample_interval=3e-06, upsample=False)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        cam.zernike_coeffs_train.fill_(-0.6)
        cam.zernike_coeffs_no_train2.copy_((torch.rand(11, 1, 1, generator=g) - 0.5) * 0.4)
    img = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    w = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    sensor, psf, coeffs, loss, g_sensor, _ = run(cam, img, None, 1, w)
    torch.manual_seed(1)
    noise = torch.rand([1, 112, 112, 1])
    # stage-level outputs from the reference's own functions
    psfs = psf.detach().permute(1, 2, 0, 3)
    raw = RU.img_psf_conv(img, psfs)
    otf = RU.psf2otf(psfs, output_size=[64, 64])
    delta_psf = torch.zeros(32, 32, 1, 3)
    delta_psf[16, 16] = 1.0
    raw_delta = RU.img_psf_conv(img, delta_psf)
    np.savez_compressed(os.path.join(HERE, "ic_tiny.npz"),
                        img=img.numpy(), w=w.numpy(), noise_u01=noise.numpy(),
                        coeffs=coeffs.detach().numpy(), volume=cam.zernike_volume.numpy(),
                        sensor=sensor.detach().numpy(), psf=psf.detach().numpy(),
                        raw=raw.numpy(), otf=otf.numpy(), raw_delta=raw_delta.numpy(),
                        grad_sensor_w=g_sensor.numpy())
    print("ic_tiny: sensor", stats(sensor), "psf sum", psf.sum().item())

    # ---- G2: real size (train.py:64-66), coefficients of Camera/Model.pth ----------
    cam = OpticsZernike(input_shape=[None, 256, 256, 3], device=cpu, zernike_terms=350, patch_size=256,
                        height_tolerance=2e-8, sensor_distance=0.025, wave_resolution=[896, 896],
                        sample_interval=3e-06, upsample=False)
    ck = torch.load(os.path.join(REF, "Image_Caption/Camera/Model.pth"), map_location=cpu, weights_only=False)["model"]
    tr = ck["optics.zernike_coeffs_train"]
    img = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    w = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(5))
    out = {}
    for tag in ("init", "modelpth"):
        with torch.no_grad():
            if tag == "modelpth":
                cam.zernike_coeffs_no_train.copy_(ck["optics.zernike_coeffs_no_train"])
                cam.zernike_coeffs_train.copy_(tr[0].reshape(1, 1))
                cam.zernike_coeffs_no_train2.copy_(tr[1:])
        sensor, psf, coeffs, loss, g_sensor, g_loss = run(cam, img, "3", 1, w)
        out.update({
            f"{tag}_coeffs": coeffs.detach().numpy().reshape(-1),
            f"{tag}_psf": psf.detach().numpy(),                      # [1,256,256,3] f64, disk-masked
            f"{tag}_loss": np.array(loss.item()),
            f"{tag}_sensor_sub": sensor.detach()[:, :, ::8, ::8].numpy(),
            f"{tag}_sensor_edge": sensor.detach()[:, :, :3, :].numpy(),
            f"{tag}_sensor_stats": stats(sensor),
            f"{tag}_grad_sensor_w": g_sensor.numpy(),
            f"{tag}_grad_loss": g_loss.numpy(),
        })
        print("ic_real", tag, "loss", loss.item(), "sensor", stats(sensor))
    out["noise_seed"] = np.array(1)
    torch.manual_seed(1)
    out["noise_stats"] = stats(torch.rand([1, 896, 896, 1]))
    out["volume_sub"] = cam.zernike_volume[:, ::16, ::16].numpy()
    out["volume_stats"] = stats(cam.zernike_volume)
    np.savez_compressed(os.path.join(HERE, "ic_real.npz"), **out)


# --------------------------------------------------------------------------- FD
def gen_fd():
    install_standins()
    sys.path.insert(0, os.path.join(REF, "Face-DeId"))
    from Camera.Optics import Camera
    out = {}
    for n, terms in ((64, 21), (256, 300), (512, 300)):
        torch.manual_seed(3)
        cam = Camera(device="cpu", N=n, zernike_terms=terms)
        img = torch.rand(2, 3, n, n, generator=torch.Generator().manual_seed(0)) * 2 - 1
        w = torch.rand(2, 3, n, n, generator=torch.Generator().manual_seed(5))
        sensor = cam(img)
        l = (sensor * w).sum() + 1e3 * cam.loss_rad + 1e6 * cam.centering_loss
        grad, = torch.autograd.grad(l, [cam.Zer_train])
        t = f"n{n}"
        out[f"{t}_zer_train"] = cam.Zer_train.detach().numpy()
        out[f"{t}_psfs"] = cam.psfs.detach().numpy()
        out[f"{t}_loss_rad"] = np.array(cam.loss_rad.item())
        out[f"{t}_centering_loss"] = np.array(cam.centering_loss.item())
        out[f"{t}_sensor_sub"] = sensor.detach()[:, :, ::max(1, n // 32), ::max(1, n // 32)].numpy()
        out[f"{t}_sensor_stats"] = stats(sensor)
        out[f"{t}_grad"] = grad.numpy().reshape(-1)
        if n == 64:
            out[f"{t}_sensor"] = sensor.detach().numpy()
            out[f"{t}_volume"] = cam.zernike_volume.numpy()
        print("fd", n, "loss_rad", cam.loss_rad.item(), "cl", cam.centering_loss.item(), "sensor", stats(sensor),
              "psf ch sums", cam.psfs.sum((0, 2, 3)).tolist())
    np.savez_compressed(os.path.join(HERE, "fd.npz"), **out)


# --------------------------------------------------------------------------- RAFT CorrBlock
def gen_corr():
    sys.path.insert(0, os.path.join(REF, "Face-DeId"))
    from RAFT.core.corr import CorrBlock
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(1, 16, 16, 16, generator=g)
    f2 = torch.randn(1, 16, 16, 16, generator=g)
    ys, xs = torch.meshgrid(torch.arange(16), torch.arange(16), indexing="ij")
    coords = torch.stack([xs, ys], 0).float()[None] + 2.0 * torch.randn(1, 2, 16, 16, generator=g)
    blk = CorrBlock(f1, f2, num_levels=4, radius=4)
    out = blk(coords)
    np.savez_compressed(os.path.join(HERE, "corr.npz"), f1=f1.numpy(), f2=f2.numpy(), coords=coords.numpy(),
                        out=out.numpy(), pyr0=blk.corr_pyramid[0].numpy(), pyr3=blk.corr_pyramid[3].numpy())
    print("corr", out.shape, stats(out))


# --------------------------------------------------------------------------- FAN heat-map regressor
def fill_by_name(module, scale_bn=True):
    """Deterministic parameters / buffers keyed on state_dict names (no 27 MB weight fixture): every tensor is drawn
    from a generator seeded with a CRC of its name.  Used identically by the oracle / product tests."""
    import zlib
    with torch.no_grad():
        for name, t in module.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith("num_batches_tracked"):
                continue
            if name.endswith("running_var"):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif name.endswith("running_mean"):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1 and ("bn" in name or "downsample.0" in name):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75 if name.endswith("weight") else torch.randn(t.shape, generator=g) * 0.1)
            elif t.dim() == 1:
                t.copy_(torch.randn(t.shape, generator=g) * 0.05)
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5 * (0.004 if name.startswith("l0.") else 1.0))


def gen_fan():
    install_standins()
    munch = types.ModuleType("munch")
    munch.Munch = dict
    sk = types.ModuleType("skimage")
    sk.filters = types.ModuleType("skimage.filters")
    sk.filters.gaussian = lambda *a, **k: None
    for name, mod in [("munch", munch), ("skimage", sk), ("skimage.filters", sk.filters)]:
        sys.modules[name] = mod
    sys.path.insert(0, os.path.join(REF, "Face-DeId"))
    from core.wing import FAN
    fan = FAN().eval()
    fill_by_name(fan)
    out = {}
    for tag, shape in (("b2_256", (2, 3, 256, 256)), ("b1_512", (1, 3, 512, 512))):
        x = torch.rand(shape, generator=torch.Generator().manual_seed(0)) * 2 - 1
        with torch.no_grad():
            hm = fan.get_heatmap(x, Privacy=True)
            xr = F.interpolate(x, size=256, mode="bilinear") * 0.5 + 0.5
            raw = fan(xr)[0][-1]
        out[f"{tag}_hm0"], out[f"{tag}_hm1"] = hm[0].numpy(), hm[1].numpy()
        out[f"{tag}_raw_sub"] = raw[:, ::7, ::4, ::4].numpy()
        out[f"{tag}_raw_stats"] = stats(raw)
        print("fan", tag, "raw", stats(raw), "hm0", stats(hm[0]), "hm1", stats(hm[1]))
    out["state_names"] = np.array(sorted(fan.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, "fan.npz"), **out)


def _face_deid_standins():
    install_standins()
    munch = types.ModuleType("munch")
    munch.Munch = dict
    sk = types.ModuleType("skimage")
    sk.filters = types.ModuleType("skimage.filters")
    sk.filters.gaussian = lambda *a, **k: None
    for name, mod in [("munch", munch), ("skimage", sk), ("skimage.filters", sk.filters)]:
        sys.modules[name] = mod
    if os.path.join(REF, "Face-DeId") not in sys.path:
        sys.path.insert(0, os.path.join(REF, "Face-DeId"))


def _fill_plain(module):
    with torch.no_grad():
        for name, t in module.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if t.dim() > 1:
                t.copy_(torch.randn(t.shape, generator=g) * (1.0 / t[0].numel()) ** 0.5)
            elif name.endswith("weight"):
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            else:
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)


def gen_fan_train():
    """FAN.get_heatmap_train (core/wing.py:262-272): the landmark heat-maps WITH autograd; gradient w.r.t. the input image."""
    _face_deid_standins()
    from core.wing import FAN
    fan = FAN().eval()
    fill_by_name(fan)
    x = (torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(5)) * 2 - 1).requires_grad_(True)
    hm = fan.get_heatmap_train(x, Privacy=True)
    w0 = torch.rand(hm[0].shape, generator=torch.Generator().manual_seed(6))
    w1 = torch.rand(hm[1].shape, generator=torch.Generator().manual_seed(7))
    ((hm[0] * w0).sum() + (hm[1] * w1).sum()).backward()
    np.savez_compressed(os.path.join(HERE, "fan_train.npz"), x=x.detach().numpy(), hm0=hm[0].detach().numpy(), hm1=hm[1].detach().numpy(),
                        gx=x.grad.numpy())
    print("fan_train hm0", stats(hm[0].detach()), "gx", stats(x.grad))


def gen_stargan():
    """StarGAN-v2 blocks (core/model.py:12-124): ResBlk (normalised, down-sampling, learned shortcut), AdainResBlk (up-sampling,
    learned shortcut): outputs, input gradients and every parameter gradient."""
    _face_deid_standins()
    from core.model import ResBlk, AdainResBlk
    out = {}
    cases = {"res": (ResBlk(64, 128, normalize=True, downsample=True), (2, 64, 16, 16), None),
             "res_plain": (ResBlk(128, 128, normalize=False, downsample=False), (1, 128, 8, 12), None),
             "ada": (AdainResBlk(128, 64, style_dim=64, w_hpf=0, upsample=True), (2, 128, 8, 8), (2, 64))}
    for tag, (m, xs, ss) in cases.items():
        _fill_plain(m)
        x = torch.randn(xs, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
        args = (x,)
        if ss is not None:
            s = torch.randn(ss, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
            args = (x, s)
        y = m(*args)
        w = torch.rand(y.shape, generator=torch.Generator().manual_seed(3))
        (y * w).sum().backward()
        out[f"{tag}_x"], out[f"{tag}_y"], out[f"{tag}_gx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()
        if ss is not None:
            out[f"{tag}_s"], out[f"{tag}_gs"] = s.detach().numpy(), s.grad.numpy()
        for n, p in m.named_parameters():
            out[f"{tag}_g_{n}"] = p.grad.numpy()
        print("stargan", tag, stats(y.detach()), stats(x.grad))
    np.savez_compressed(os.path.join(HERE, "stargan.npz"), **out)


# --------------------------------------------------------------------------- attention decoder (Image_Caption/models.py)
def gen_decoder():
    install_standins()
    sys.path.insert(0, os.path.join(REF, "Image_Caption"))
    import models as ref_models
    from torch.nn.utils.rnn import pack_padded_sequence
    ref_models.device = torch.device("cpu")                   # models.py:5 hard-wires cuda:0
    g = torch.Generator().manual_seed(0)
    B, S, E, A, M, D, V, L = 5, 6, 128, 128, 24, 40, 50, 11   # E, A: multiples of 128 (the HIP path's MFMA tiles)
    dec = ref_models.DecoderWithAttention(attention_dim=A, embed_dim=M, decoder_dim=D, vocab_size=V, encoder_dim=E, dropout=0.3).eval()
    fill_by_name(dec)
    with torch.no_grad():
        dec.embedding.weight.copy_(torch.rand(V, M, generator=g) * 0.2 - 0.1)
    enc = torch.randn(B, S, S, E, generator=g).requires_grad_(True)
    caps = torch.randint(0, V, (B, L), generator=g)
    caplens = torch.tensor([[7], [11], [4], [9], [7]])
    preds, caps_sorted, dec_len, alphas, order = dec(enc, caps, caplens)
    targets = caps_sorted[:, 1:]
    sc = pack_padded_sequence(preds, dec_len, batch_first=True)
    tg = pack_padded_sequence(targets, dec_len, batch_first=True)
    loss = torch.nn.functional.cross_entropy(sc.data, tg.data) + 1.0 * ((1.0 - alphas.sum(dim=1)) ** 2).mean()   # train.py:276-282
    loss.backward()
    out = {"enc": enc.detach().numpy(), "caps": caps.numpy(), "caplens": caplens.numpy(), "preds": preds.detach().numpy(),
           "alphas": alphas.detach().numpy(), "order": order.numpy(), "dec_len": np.array(dec_len), "loss": loss.item(),
           "g_enc": enc.grad.numpy(), "emb_weight": dec.embedding.weight.detach().numpy(), "dims": np.array([B, S, E, A, M, D, V, L])}
    for n, p_ in dec.named_parameters():
        out["g_" + n] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "decoder.npz"), **out)
    print("decoder loss", loss.item(), "preds", stats(preds), "alphas", stats(alphas), "g_enc", stats(enc.grad))



# --------------------------------------------------------------------------- ResNet-101 encoder (Image_Caption/models.py:8-54)
def gen_encoder():
    """The reference's own ``models.Encoder`` (models.py:8-54: children()[:-2] of torchvision.models.resnet101, fine_tune() on
    children [5:], AdaptiveAvgPool2d + permute) run on CPU in train mode (train.py:245).  torchvision is absent offline: the stand-in
    for ``torchvision.models.resnet101`` returns a module with torchvision's child order (conv1, bn1, relu, maxpool, layer1-4,
    avgpool, fc) whose trunk is assembled from oracle/resnet.py's pieces (that restatement is itself pinned against the independent
    ResNet-101 v1.5 of `transformers`, tests/test_oracle_trunk_pin.py); everything the Encoder class does with it is the
    reference's code.  Weights: tests/trunk_fill.py (by state_dict name).  4 x 3 x 64 x 64 input: layer 4 sees a 2 x 2 map."""
    install_standins()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import resnet as R
    from trunk_fill import fill_trunk_by_name
    from torch import nn

    class TVResNet(nn.Module):                                  # torchvision.models.resnet.ResNet's attribute / child order
        def __init__(self):
            super().__init__()
            t = R.make_resnet101_trunk()
            self.conv1, self.bn1, self.relu, self.maxpool = t[0], t[1], t[2], t[3]
            self.layer1, self.layer2, self.layer3, self.layer4 = t[4], t[5], t[6], t[7]
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(2048, 1000)

    sys.modules["torchvision"].models.resnet101 = lambda pretrained=False, **kw: TVResNet()
    sys.path.insert(0, os.path.join(REF, "Image_Caption"))
    import models as ref_models
    out = {}
    for tag, E in (("e3", 3), ("e36", 36)):
        enc = ref_models.Encoder(encoded_image_size=E)
        fill_trunk_by_name(enc)
        enc.train()
        img = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
        y = enc(img)
        w = torch.rand(y.shape, generator=torch.Generator().manual_seed(5))
        (y * w).sum().backward()
        names = [n for n, _ in enc.named_parameters()]
        if tag == "e3":
            out["state_names"] = np.array(list(enc.state_dict().keys()))
            out["param_names"] = np.array(names)
            out["requires_grad"] = np.array([p.requires_grad for _, p in enc.named_parameters()])
            out["img_grad"] = img.grad.numpy()
            out["out"] = y.detach().numpy()
            # every parameter gradient as (sum, sum of squares, first 8 values); None (frozen) -> zeros
            gs = np.zeros((len(names), 10))
            for i, (_, p) in enumerate(enc.named_parameters()):
                if p.grad is not None:
                    g = p.grad.double().reshape(-1)
                    gs[i, 0], gs[i, 1] = g.sum().item(), (g * g).sum().item()
                    gs[i, 2:2 + min(8, g.numel())] = g[:8].numpy()
            out["param_grad_stats"] = gs
            sd = enc.state_dict()
            out["running_mean"] = np.concatenate([sd[k].numpy() for k in sd if k.endswith("running_mean")])
            out["running_var"] = np.concatenate([sd[k].numpy() for k in sd if k.endswith("running_var")])
            out["num_batches_tracked"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")])
            out["g_layer2_conv1"] = enc.resnet[5][0].conv1.weight.grad.numpy()          # one whole weight gradient (128x256x1x1)
            out["g_layer4_bn3"] = enc.resnet[7][2].bn3.weight.grad.numpy()
        else:
            out["out36_sub"] = y.detach()[:, ::7, ::7, ::16].numpy()
            out["out36_stats"] = stats(y)
        print("encoder", tag, y.shape, stats(y), "img grad", stats(img.grad))
    np.savez_compressed(os.path.join(HERE, "encoder.npz"), **out)

# --------------------------------------------------------------------------- RAFT SepConvGRU
def gen_raft_gru():
    """RAFT's SepConvGRU (RAFT/core/update.py:33-60): the reference module with parameters filled by name, three chained updates."""
    sys.path.insert(0, os.path.join(REF, "Face-DeId"))
    from RAFT.core.update import SepConvGRU
    torch.manual_seed(0)
    gru = SepConvGRU(hidden_dim=128, input_dim=256).eval()
    with torch.no_grad():
        for name, t in gru.state_dict().items():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            t.copy_(torch.randn(t.shape, generator=g) * ((1.0 / t[0].numel()) ** 0.5 if t.dim() > 1 else 0.1))
    h = torch.tanh(torch.randn(2, 128, 12, 20, generator=torch.Generator().manual_seed(1)))
    x = torch.randn(2, 256, 12, 20, generator=torch.Generator().manual_seed(2))
    outs = []
    with torch.no_grad():
        hh = h
        for _ in range(3):
            hh = gru(hh, x)
            outs.append(hh.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "raft_gru.npz"), h=h.numpy(), x=x.numpy(), out1=outs[0], out3=outs[2])
    print("raft_gru", outs[2].shape, stats(torch.from_numpy(outs[2])))


# --------------------------------------------------------------------------- SSIM loss (Image_Caption/pytorch_ssim)
def gen_ssim():
    sys.path.insert(0, os.path.join(REF, "Image_Caption"))
    import pytorch_ssim
    out = {}
    for tag, shape in (("a", (2, 3, 64, 64)), ("b", (3, 3, 40, 52))):
        g = torch.Generator().manual_seed(7)
        x = torch.rand(shape, generator=g)
        y = (x + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
        x1, y1 = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        v = pytorch_ssim.SSIM()(x1, y1)                        # window 11, sigma 1.5, size_average=True
        v.backward()
        per = pytorch_ssim.ssim(x, y, size_average=False)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_y": y.numpy(), f"{tag}_ssim": v.item(), f"{tag}_gx": x1.grad.numpy(),
                    f"{tag}_gy": y1.grad.numpy(), f"{tag}_per_image": per.numpy()})
        print("ssim", tag, v.item(), per.tolist())
    np.savez_compressed(os.path.join(HERE, "ssim.npz"), **out)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "all":
        for w in ("ic", "fd", "corr", "fan", "decoder", "ssim", "raft_gru", "fan_train", "stargan", "encoder"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
    else:
        {"ic": gen_ic, "fd": gen_fd, "corr": gen_corr, "fan": gen_fan, "decoder": gen_decoder, "ssim": gen_ssim, "raft_gru": gen_raft_gru, "fan_train": gen_fan_train, "stargan": gen_stargan, "encoder": gen_encoder}[what]()
