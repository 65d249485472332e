"""Which torch (aten) operators put kernels into the headline step, with device time and input shapes: torch.profiler over 3 steps.
Run on the GPU box: python tools/torch_ops_in_step.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
step, _ = bench.make_step(camera, encoder, 128, dev, None, None)
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    dt = getattr(ev, "self_device_time_total", None)
    if dt is None:
        dt = ev.self_cuda_time_total
    if dt / N >= 8 and ev.key.startswith("aten::"):
        frames = [f for f in ev.stack if ("ppv" in f or "privacy" in f or "bench.py" in f or "optim" in f)]
        rows.append((dt / N, ev.count / N, ev.key, str(ev.input_shapes)[:70], (frames[0] if frames else "?")[-90:]))
for dt, c, k, shp, where in sorted(rows, reverse=True)[:40]:
    print(f"{dt:8.1f} us/step {c:5.1f}x  {k:28s} {shp:70s} {where}")
