"""Phase timeline of conv1x1_stream_kernel from a diagnostic build (csrc/build_stamps.sh, -DPPV_STAMPS), next to back-to-back
launch times of the streaming and the tiled kernel on the same cold operands.
Run:  PPV_LIB_PATH=privacy-preserving-vision_amd/lib_stamps/libppv_hip.so python tools/conv_timeline.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ppv_amd.convops as co

B = 128
lib = co.L()
have_stamps = hasattr(lib, "ppv_debug_set_stamps")
if have_stamps:
    lib.ppv_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda")


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


def timeline(name, fn):
    if not have_stamps:
        return
    buf.zero_()
    lib.ppv_debug_set_stamps(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.ppv_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 0] > 0]
    if not len(s):
        print(f"  {name}: no stamps (tiled kernel)")
        return
    s = (s - s[:, 0].min()) * 0.01                                    # us since the first workgroup's entry (100 MHz counter)
    c, l = s[:, :8], s[:, 8:]
    md = lambda v: f"{np.median(v):5.2f}/{np.percentile(v, 90):5.2f}"
    print(f"  {name}: {len(s)} WGs, starts p50 {np.median(c[:, 0]):5.1f} max {c[:, 0].max():5.1f}, last end {c[:, 7].max():5.1f} us | consumer wave 0 "
          f"(median/p90 us): rows requested {md(c[:, 1] - c[:, 0])}, first barrier {md(c[:, 2] - c[:, 1])}, first MFMAs done {md(c[:, 3] - c[:, 2])}, "
          f"first epilogue {md(c[:, 4] - c[:, 3])}, second step {md(c[:, 5] - c[:, 4])}, remaining steps {md(c[:, 6] - c[:, 5])}, fold {md(c[:, 7] - c[:, 6])}, "
          f"total {md(c[:, 7] - c[:, 0])} | loader wave: two tiles issued {md(l[:, 1] - c[:, 0])}, tile 0 landed {md(l[:, 2] - l[:, 1])}, "
          f"barrier 0 wait {md(l[:, 3] - l[:, 2])}, tile 1 landed {md(l[:, 4] - l[:, 3])}, barrier 1 wait {md(l[:, 5] - l[:, 4])}")


def cold(shape, n=6):
    return [torch.randn(*shape, device="cuda").bfloat16() for _ in range(n)]


for cin, cout, h in [(256, 1024, 16), (128, 512, 32), (64, 256, 64)]:
    k = 1
    xs = cold((B, h, h, cin))
    w = co.weight_layout(torch.randn(cout, cin, k, k, device="cuda") * 0.05, 0)
    M = B * h * h
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    it = [0]

    def fwd():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % len(xs)], w, 1, 0, stat_part=part)
    gs = cold((B, h, h, cin))
    wd = co.weight_layout(torch.randn(cin, cout, k, k, device="cuda") * 0.05, 1)       # conv(cout -> cin): its data gradient is cin -> cout
    xr = cold((B, h, h, cout), 3)
    add = cold((B, h, h, cout), 3)
    pr = torch.zeros(64 * cout, device="cuda")
    bits = torch.randint(0, 255, (M * cout // 8,), dtype=torch.uint8, device="cuda")

    def dg():
        it[0] += 1
        return co.conv_dgrad(gs[it[0] % len(gs)], wd, 1, 0, (h, h), addend=add[it[0] % 3], relu_bits=bits, red=(xr[it[0] % 3], pr))
    for nm, fn, byts in (("fwd", fwd, M * (cin + cout) * 2), ("dgrad+addend+mask+sums", dg, M * (cin + 3 * cout) * 2 + M * cout // 8)):
        res = {}
        for v in (0, 7):
            lib.ppv_conv_set_variant(v)
            res[v] = timed(fn)
        lib.ppv_conv_set_variant(0)
        print(f"{nm} {cin}->{cout} h{h}: streaming {res[0]:6.1f} us ({byts/res[0]/1e6:4.2f} TB/s), tiled {res[7]:6.1f} us ({byts/res[7]/1e6:4.2f} TB/s); HBM floor at 6 TB/s {byts/6e6:5.1f} us")
        timeline(nm, fn)
    del xs, gs, xr, add
