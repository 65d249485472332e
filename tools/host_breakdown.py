"""Host time of the bench step by section (perf_counter, no profiler, nothing synchronises inside): camera forward, encoder forward,
head + losses, backward, optimisers.  Mirrors bench.make_step's body."""
import os, sys, time, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
enc_params = [p for p in encoder.parameters() if p.requires_grad]
cam_params = [p for p in camera.parameters() if p.requires_grad]
opt_enc = torch.optim.Adam(enc_params, lr=1e-4, fused=True)
opt_cam = torch.optim.Adam(cam_params, lr=5e-7)
imgs = torch.rand(128, 3, 256, 256, generator=torch.Generator().manual_seed(0)).to(dev)
T = {k: 0.0 for k in ("camera", "encoder", "loss", "zero_grad", "backward", "opt_cam", "opt_enc")}
def step(acc):
    t = [time.perf_counter()]
    sensor, psf, coeffs, loss_psf = camera(imgs, None, "3"); t.append(time.perf_counter())
    enc_out = encoder(sensor); t.append(time.perf_counter())
    loss = 0.4 * bench.head_stand_in(enc_out) + 6 * (1 - torch.nn.functional.mse_loss(imgs, sensor)) + 30 * loss_psf; t.append(time.perf_counter())
    opt_enc.zero_grad(set_to_none=True); opt_cam.zero_grad(set_to_none=True); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt_cam.step(); camera.zernike_coeffs_train[1:].data.clamp_(-1, 1); t.append(time.perf_counter())
    grads = [p.grad for p in enc_params]
    torch._foreach_clamp_min_(grads, -5.0); torch._foreach_clamp_max_(grads, 5.0)
    opt_enc.step(); encoder.prefetch_weight_layouts(); t.append(time.perf_counter())
    if acc:
        for k, a, b in zip(T, t, t[1:]): T[k] += b - a
for _ in range(4): step(False)
torch.cuda.synchronize()
N = int(os.environ.get('HB_STEPS', '10'))
for _ in range(N): step(True)
torch.cuda.synchronize()
print("  ".join(f"{k} {v / N * 1e3:.2f} ms" for k, v in T.items()), " total %.2f ms" % (sum(T.values()) / N * 1e3))
