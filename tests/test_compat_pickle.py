"""Drop-in plumbing the reference relies on besides forward(): the bare import paths of its scripts (compat/ shims) and
whole-module pickling of its checkpoints (Image_Caption/utils.py:387-395, train.py:139-151)."""
import io
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compat_shims_resolve_to_the_mi355x_classes():
    code = ("import models, Camera.Lens, pytorch_ssim, ppv_amd.encoder, ppv_amd.decoder, ppv_amd.camera_lens, ppv_amd.ssim;"
            "assert models.Encoder is ppv_amd.encoder.Encoder and models.DecoderWithAttention is ppv_amd.decoder.DecoderWithAttention;"
            "assert Camera.Lens.OpticsZernike is ppv_amd.camera_lens.OpticsZernike and pytorch_ssim.SSIM is ppv_amd.ssim.SSIM;"
            "print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "compat", "image_caption")]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
    code = "import Camera.Optics, ppv_amd.camera_optics; assert Camera.Optics.Camera is ppv_amd.camera_optics.Camera; print('ok')"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "compat", "face_deid")]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
def test_whole_module_pickle_round_trip():
    from ppv_amd.encoder import Encoder
    from ppv_amd.decoder import DecoderWithAttention
    from ppv_amd.camera_lens import OpticsZernike
    from ppv_amd.camera_optics import Camera
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    enc = Encoder(encoded_image_size=9, layers=(1, 1, 1, 1)).cuda().train()
    dec = DecoderWithAttention(128, 32, 48, 30, encoder_dim=2048, dropout=0.0).cuda().eval()
    cam = OpticsZernike(input_shape=[None, 128, 128, 3], device=dev, zernike_terms=36, patch_size=128, height_tolerance=2e-8,
                        sensor_distance=0.025, wave_resolution=[448, 448], sample_interval=3e-06, upsample=False)
    fd = Camera(device=dev, N=256, zernike_terms=50)
    img = torch.rand(2, 3, 64, 64, device="cuda")
    enc(img).sum().backward()                                  # populate the runtime state that must NOT be pickled (streams, layouts)
    buf = io.BytesIO()
    torch.save({"encoder": enc, "decoder": dec, "camera": cam, "fd": fd}, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    enc.eval(); back["encoder"].eval()
    with torch.no_grad():
        assert torch.equal(back["encoder"](img), enc(img))
        caps, lens = torch.randint(0, 30, (2, 6), device="cuda"), torch.tensor([[6], [4]], device="cuda")
        # the decoder's f32 GEMM splits K over workgroups and adds the partials atomically: same values, summation order free
        torch.testing.assert_close(back["decoder"](enc(img), caps, lens)[0], dec(enc(img), caps, lens)[0], rtol=1e-5, atol=1e-6)
        x = torch.rand(2, 3, 128, 128, device="cuda")
        noise = torch.rand(1, 448, 448, 1, device="cuda")
        a = cam(x, None, None, noise_u01=noise)[0]
        b = back["camera"](x, None, None, noise_u01=noise)[0]
        assert torch.equal(a, b)
        y = torch.rand(2, 3, 256, 256, device="cuda")
        assert torch.equal(back["fd"](y), fd(y))
    back["encoder"].train()
    back["encoder"](img).sum().backward()                      # the reloaded module trains
