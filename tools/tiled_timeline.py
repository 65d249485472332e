"""Phase timeline of the tiled conv kernel (conv_gemm_pipe_kernel) from the diagnostic build (csrc/build_stamps.sh): where a workgroup's
lifetime goes -- ring fill, K loop, epilogue, store drain -- for the trunk's layer-2/3 shapes at B = 128.
Run:  PPV_LIB_PATH=privacy-preserving-vision_amd/lib_stamps/libppv_hip.so python tools/tiled_timeline.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ppv_amd.convops as co

B = 128
lib = co.L()
lib.ppv_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 16384, dtype=torch.int64, device="cuda")


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def timeline(fn):
    buf.zero_()
    lib.ppv_debug_set_stamps(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.ppv_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 0] > 0]
    if not len(s):
        print("   no stamps")
        return
    s = (s - s[:, 0].min()) * 0.01
    md = lambda v: f"{np.median(v):5.2f}/{np.percentile(v, 90):5.2f}"
    st = np.sort(s[:, 0])
    print(f"   {len(s)} WGs; starts: p25 {st[len(st)//4]:5.1f} p50 {st[len(st)//2]:5.1f} p75 {st[3*len(st)//4]:5.1f} max {st[-1]:5.1f}; last drain {s[:, 6].max():5.1f} us\n"
          f"   per WG median/p90 us: address setup {md(s[:, 5] - s[:, 0])}, ring prologue issued {md(s[:, 1] - s[:, 5])}, first stage lands {md(s[:, 2] - s[:, 1])}, "
          f"K loop {md(s[:, 3] - s[:, 2])}, epilogue: tile -> LDS {md(s[:, 7] - s[:, 3])}, barrier + stat atomics + store loop {md(s[:, 4] - s[:, 7])}, store drain {md(s[:, 6] - s[:, 4])}, total {md(s[:, 6] - s[:, 0])}")


def cold(shape, n=6):
    return [torch.randn(*shape, device="cuda").bfloat16() for _ in range(n)]


lib.ppv_conv_set_variant(7)                                  # tiled kernels only
for cin, cout, h, k in [(1024, 256, 16, 1), (256, 1024, 16, 1), (256, 256, 16, 3), (512, 128, 32, 1), (128, 512, 32, 1), (128, 128, 32, 3),
                        (256, 64, 64, 1), (64, 256, 64, 1)]:
    xs = cold((B, h, h, cin))
    w = co.weight_layout(torch.randn(cout, cin, k, k, device="cuda") * 0.05, 0)
    M = B * h * h
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    it = [0]

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % len(xs)], w, 1, k // 2, stat_part=part)
    t = timed(fwd)
    byts = M * (cin + cout) * 2
    print(f"fwd {cin}->{cout} {k}x{k} h{h}: {t:6.1f} us ({byts / t / 1e6:4.2f} TB/s algorithmic, {2 * M * cin * cout * k * k / t / 1e6:5.0f} TF/s)")
    timeline(fwd)
    del xs
lib.ppv_conv_set_variant(0)
