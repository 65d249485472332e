"""Is the step host-bound?  Times the Python enqueue of K steps (no synchronisation inside) against the wall time to completion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
step, _ = bench.make_step(camera, encoder, 128, dev, None)
for _ in range(3):
    step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step, complete {1e3*(t2-t0)/K:.2f} ms/step, tail after last enqueue {1e3*(t2-t1):.2f} ms")
