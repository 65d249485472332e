"""Where the host's 19 ms of enqueue per step go: cProfile of 5 steps (device idle time does not matter: nothing synchronises)."""
import os, sys, cProfile, pstats
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
torch.autograd.set_multithreading_enabled(False)      # backward on the calling thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
