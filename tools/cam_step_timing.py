"""Why does tools/bench_camera.py report 4 ms for the IC step on some runs?  Same step, timed in repeated windows of 10 from the start."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd
from ppv_amd.camera_lens import OpticsZernike
dev = torch.device("cuda", 0)
B = 64
img = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
cam = OpticsZernike(input_shape=[None, 256, 256, 3], device=dev, zernike_terms=350, patch_size=256, height_tolerance=2e-8,
                    sensor_distance=0.025, wave_resolution=[896, 896], sample_interval=3e-06, coeff_layout="B")
def ic_step():
    cam.zernike_coeffs_train.grad = None
    sensor, psf, coeffs, loss = cam(img, None, "3")
    (sensor.mean() + loss).backward()
for i in range(3): ic_step()
torch.cuda.synchronize()
for rep in range(6):
    ts = []
    t0 = time.perf_counter()
    for i in range(10):
        a = time.perf_counter(); ic_step(); ts.append(round((time.perf_counter() - a) * 1e3, 2))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print('window', rep, 'enqueue ms/step', round((t1 - t0) * 100, 3), 'total ms/step', round((t2 - t0) * 100, 3), 'device allocs', st['num_device_alloc'],
          'reserved MB', st['reserved_bytes.all.current'] >> 20, ts)
