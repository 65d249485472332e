"""BASELINE.json config 2: camera alone, 64 x 3 x 256 x 256 fp32, forward + backward, one MI355X.
Reports images/sec for the IC OpticsZernike (896/350, prueba '3', loss = sensor.mean() + loss_psf) and the FD Camera
(N = 256, 300 terms, forward), and the achieved HBM rate of the FFT-convolution kernels against their algorithmic bytes
(2.5 MB per 512^2 plane forward)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd  # noqa: F401
import ppv_amd.fftconv as fc
from ppv_amd.camera_lens import OpticsZernike
from ppv_amd.camera_optics import Camera

dev = torch.device("cuda", 0)
B = 64
img = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
cam = OpticsZernike(input_shape=[None, 256, 256, 3], device=dev, zernike_terms=350, patch_size=256, height_tolerance=2e-8,
                    sensor_distance=0.025, wave_resolution=[896, 896], sample_interval=3e-06, coeff_layout="B")


def ic_step():
    cam.zernike_coeffs_train.grad = None
    sensor, psf, coeffs, loss = cam(img, None, "3")
    (sensor.mean() + loss).backward()


def timeit(fn, n=10, w=3, windows=3):
    """Median of `windows` timed windows of n calls after w warm-up calls.  (One window right after the warm-up measured 4-10 ms per IC
    step on some boxes against 0.93 ms in every later window: round 5, tools/cam_step_timing.py.)"""
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[len(ts) // 2]


t_ic = timeit(ic_step)
fd = Camera(device=dev, N=256, zernike_terms=300)
imgn = img * 2 - 1
t_fd = timeit(lambda: fd(imgn))
# FFT convolution alone (forward): algorithmic bytes 2.5 MB per 512^2 plane
psf = torch.rand(3, 256, 256, device=dev)
otf = fc.otf_build(psf, 256, 512)
ws = torch.empty(ppv_amd._lib.lib().ppv_fftconv_workspace_bytes(B, 3, 512), dtype=torch.uint8, device=dev)
t_conv = timeit(lambda: fc.fftconv_fwd(img, otf, 0, workspace=ws), n=20)
alg = B * 3 * 2.5e6


def cpu_baseline():
    """the reference CPU path of this configuration (oracle/ic_camera.py: torch-CPU restatement of Lens.py / Utils.py,
    prueba '3', loss = sensor.mean() + loss_psf) on a bounded sample -- 16 of the 64 images, one warm-up + three timed passes, median."""
    from oracle import ic_camera as ic
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    Bc = 16
    vol = cam.zernike_volume.cpu()
    coeffs = cam._concat().detach().cpu().requires_grad_(True)
    m1, m2 = ic.disk_masks()
    im = img[:Bc].cpu()
    noise = torch.rand(1, 896, 896, 1, generator=torch.Generator().manual_seed(1))

    def once():
        coeffs.grad = None
        sensor, psf_, lpsf = ic.forward(im, coeffs, vol, noise, prueba="3", mask_1=m1, mask_2=m2, height_tolerance=2e-8,
                                        sensor_distance=0.025, sample_interval=3e-6)
        (sensor.mean() + lpsf).backward()

    t0 = time.perf_counter(); once(); warm = time.perf_counter() - t0
    ts = [warm]
    if warm < 12:
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); once(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": round(Bc / t, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"IC camera (896/350/256, prueba '3') fwd+bwd fp32/fp64 on the CPU oracle, B={Bc} of 64 @256x256, "
                      + (f"1 warm-up + {len(ts)} timed passes, median" if len(ts) > 1 else "single pass"),
            "passes_s": [round(x, 3) for x in ts]}


print(json.dumps({
    "config": "camera alone 64x3x256x256 fp32 (BASELINE.json configs[1])",
    "ic_fwd_bwd_images_per_s": round(B / t_ic, 1), "ic_ms": round(t_ic * 1e3, 3),
    "fd_fwd_images_per_s": round(B / t_fd, 1), "fd_ms": round(t_fd * 1e3, 3),
    "fftconv_fwd_ms": round(t_conv * 1e3, 3), "fftconv_algorithmic_GBps": round(alg / t_conv / 1e9, 1),
    "fftconv_frac_of_8TBps": round(alg / t_conv / 8e12, 3),
    "cpu_baseline": None if os.environ.get("PPV_NO_CPU_BASELINE") else cpu_baseline()}))
