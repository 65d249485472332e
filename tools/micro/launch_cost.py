"""Host cost of one kernel launch through the C ABI (ctypes crossing + hipLaunchKernel), no synchronisation: bn_finalize (tiny kernel)
called 4000 times; then the same through torch (an elementwise add_) for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ppv_amd
from ppv_amd import _lib
from ppv_amd._lib import ptr, stream_ptr
L = _lib.lib()
dev = torch.device("cuda", 0)
C = 256
part = torch.zeros(2, 2, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); coef = torch.empty(4, C, device=dev)
args = (ptr(part), 2, 1000.0, ptr(g), ptr(b), None, None, 0.1, 1e-5, ptr(coef), C)
for rep in range(3):
    torch.cuda.synchronize()
    N = 4000
    sp = stream_ptr()
    t0 = time.perf_counter()
    for _ in range(N):
        L.ppv_bn_finalize(*args, sp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"ppv_bn_finalize via ctypes: {1e6 * (t1 - t0) / N:.2f} us per call enqueued, {1e6 * (t2 - t0) / N:.2f} us per call completed")
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4000):
    x.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"torch add_: {1e6 * (t1 - t0) / 4000:.2f} us per call enqueued")
