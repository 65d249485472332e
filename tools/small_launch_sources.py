"""Which Python lines issue the step's small torch launches (aten::fill_ / zero_ / copy_ / zeros ...)?  torch.profiler with stacks over 3 steps."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
decoder = None
if "--decoder" in sys.argv:
    from ppv_amd.decoder import DecoderWithAttention
    torch.manual_seed(3)
    decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(dev).train()
step, _ = bench.make_step(camera, encoder, 128, dev, None, decoder)
for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
N = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        step()
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::cat", "aten::contiguous", "aten::clone")
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in want:
        frames = [f for f in ev.stack if ("ppv" in f or "privacy" in f or "bench.py" in f or "optim" in f or "camera" in f or "decoder" in f)]
        rows.append((ev.count / N, ev.key, (frames[0] if frames else (ev.stack[0] if ev.stack else "?"))[-140:]))
for c, name, where in sorted(rows, reverse=True)[:60]:
    print(f"{c:7.1f}/step  {name:14s} {where}")
