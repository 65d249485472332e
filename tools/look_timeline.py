"""Phase timeline (diagnostic build with clock stamps, csrc/build_stamps.sh) of the one-round 256 x 128 tile in its plain (variant 3) and
look-ahead (variant 6) forms on the layer-3 1x1 shapes at B = 128; run once per PPV_CONV_DEBUG mode (0 whole, 1 loads only, 2 K loop only):
  PPV_LIB_PATH=privacy-preserving-vision_amd/lib_stamps/libppv_hip.so PPV_CONV_DEBUG=0 python tools/look_timeline.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ppv_amd.convops as co

B = 128
lib = co.L()
lib.ppv_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 16384, dtype=torch.int64, device="cuda")


def timeline(fn):
    buf.zero_()
    lib.ppv_debug_set_stamps(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.ppv_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 0] > 0]
    if not len(s):
        return "no stamps"
    s = (s - s[:, 0].min()) * 0.01
    md = lambda v: f"{np.median(v):5.2f}/{np.percentile(v, 90):5.2f}"
    return (f"{len(s)} WGs, last drain {s[:, 6].max():5.1f} us | setup {md(s[:, 5] - s[:, 0])} prologue issue {md(s[:, 1] - s[:, 5])} first stage lands {md(s[:, 2] - s[:, 1])} "
            f"K loop {md(s[:, 3] - s[:, 2])} tile->LDS {md(s[:, 7] - s[:, 3])} stats+store {md(s[:, 4] - s[:, 7])} drain {md(s[:, 6] - s[:, 4])} total {md(s[:, 6] - s[:, 0])}")


for cin, cout in [(1024, 256), (256, 1024)]:
    xs = [torch.randn(B, 16, 16, cin, device="cuda").bfloat16() for _ in range(6)]
    w = co.weight_layout(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, 0)
    part = torch.zeros(co.stat_tiles(B * 256), 2, cout, device="cuda")
    it = [0]

    def f():
        it[0] += 1
        return co.conv_fwd(xs[it[0] % 6], w, 1, 0, stat_part=part)
    for v in (3, 6, 4):
        lib.ppv_conv_set_variant(v)
        for _ in range(6):
            f()
        print(f"{cin}->{cout} variant {v} debug {os.environ.get('PPV_CONV_DEBUG', '0')}: {timeline(f)}", flush=True)
    lib.ppv_conv_set_variant(0)
