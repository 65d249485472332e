"""Per-step device time of the bench step from process start (events, read back at the end): does a fresh box / fresh process need
more than the driver's 5 warm-up steps to reach its steady state?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
t_import = time.perf_counter()
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
step, _ = bench.make_step(camera, encoder, 128, dev, None)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
torch.cuda.synchronize()
t0 = time.perf_counter()
ev[0].record()
for i in range(N):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
t1 = time.perf_counter()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("build %.1f s; %d steps in %.1f ms" % (t0 - t_import, N, (t1 - t0) * 1e3))
print(" ".join("%.1f" % m for m in ms))
print("mean steps 5-24: %.2f ms; mean last 10: %.2f ms" % (sum(ms[5:25]) / 20, sum(ms[-10:]) / 10))
