"""Phase timeline of conv_wgrad_stream_kernel (diagnostic build, csrc/build_stamps.sh).
Run:  PPV_LIB_PATH=privacy-preserving-vision_amd/lib_stamps/libppv_hip.so python tools/wgrad_timeline.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ppv_amd.convops as co

B = 128
lib = co.L()
lib.ppv_debug_set_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(16 * 8192, dtype=torch.int64, device="cuda")
for cin, cout, k, h in [(256, 1024, 1, 16), (1024, 256, 1, 16), (128, 512, 1, 32)]:
    xs = [torch.randn(B, h, h, cin, device="cuda").bfloat16() for _ in range(4)]
    gs = [torch.randn(B, h, h, cout, device="cuda").bfloat16() for _ in range(4)]
    acc = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
    for i in range(4):
        co.conv_wgrad(gs[i], xs[i], k, k, 1, 0, scratch=acc)
    buf.zero_()
    lib.ppv_debug_set_stamps(buf.data_ptr())
    co.conv_wgrad(gs[0], xs[0], k, k, 1, 0, scratch=acc)
    torch.cuda.synchronize()
    lib.ppv_debug_set_stamps(None)
    s = buf.cpu().numpy().reshape(-1, 16).astype(np.float64)
    s = s[s[:, 0] > 0]
    s = (s - s[:, 0].min()) * 0.01
    c, l = s[:, :8], s[:, 8:]
    md = lambda v: f"{np.median(v):5.2f}/{np.percentile(v, 90):5.2f}"
    print(f"wgrad {cin}->{cout} h{h}: {len(s)} WGs, starts p50 {np.median(c[:,0]):5.1f} max {c[:,0].max():5.1f}, last end {c[:,7].max():5.1f} us | consumer 0: "
          f"first stage landed {md(c[:,1]-c[:,0])}, 2nd {md(c[:,2]-c[:,1])}, stages 1-5 {md(c[:,3]-c[:,2])}, stages 5-13 {md(c[:,4]-c[:,3])}, rest {md(c[:,5]-c[:,4])}, "
          f"slab store {md(c[:,7]-c[:,5])}, total {md(c[:,7]-c[:,0])} | loader 0: first issue {md(l[:,1]-c[:,0])}, 5 issued {md(l[:,2]-l[:,1])}, "
          f"1st landed {md(l[:,3]-l[:,1])}, 5th landed {md(l[:,4]-l[:,3])}, 13th landed {md(l[:,5]-l[:,4])}, end {md(l[:,7]-l[:,5])}")
