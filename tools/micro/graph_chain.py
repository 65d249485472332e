"""How long does a dependent chain of tiny kernels take per kernel, eager (one stream, host ahead) against a captured hipGraph?
Decides whether graph capture of the trunk's ~800 launches per step is worth building."""
import torch
x = torch.zeros(4096, device="cuda")
big = torch.zeros(64 << 20, device="cuda")
N = 400


def chain():
    for _ in range(N):
        x.add_(1.0)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        big.add_(1.0)                      # keeps the GPU busy ~100 us so the host gets ahead of the chain
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
    return best


print(f"eager chain: {timed(chain):.2f} us per tiny kernel")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
torch.cuda.synchronize()
print(f"graph replay: {timed(g.replay):.2f} us per tiny kernel")
