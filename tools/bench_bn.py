"""Train-mode BN backward at the trunk's shapes (B = 128), cold-ish caches: the eight shapes are cycled so that a launch
never finds its own operands in L2 / MALL from the previous repetition.  A/B the fused-coefficient apply with
PPV_BN_BWD_FUSED=0/1 (read once per process)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ppv_amd  # noqa: F401,E402
from ppv_amd import convops as co  # noqa: E402

SHAPES = [(524288, 64, 2), (524288, 256, 0), (131072, 128, 2), (131072, 512, 0), (32768, 256, 2), (32768, 1024, 0),
          (8192, 512, 2), (8192, 2048, 0)]


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ops = []
    for rows, C, relu in SHAPES:
        x = torch.randn(rows, C, device=dev).bfloat16()
        gy = torch.randn(rows, C, device=dev).bfloat16()
        coef = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1,
                            torch.rand(C, device=dev) + 0.5]).contiguous()
        ops.append((x, gy, coef, relu))
    reps = 10
    tot = [0.0] * len(ops)
    for rep in range(reps + 2):
        for i, (x, gy, coef, relu) in enumerate(ops):
            part = torch.zeros(64 * x.shape[1], device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            co.bn_bwd(gy, None, x, coef, relu, part=part)
            e1.record()
            e1.synchronize()
            if rep >= 2:
                tot[i] += e0.elapsed_time(e1) * 1e3
    w = [6, 4, 8, 5, 46, 24, 6, 4]           # launches per step of each shape (t-sized BN1/BN2, 4t-sized BN3 + projections)
    step = 0.0
    for (rows, C, relu), t, k in zip(SHAPES, tot, w):
        us = t / reps
        gb = rows * C * 2 * 5 / 1e9
        step += us * k
        print(f"rows {rows:7d} C {C:5d} relu {relu}: {us:7.1f} us  ({gb / us * 1e6 / 1e3:.2f} TB/s over 5 passes)")
    print(f"weighted per step: {step / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
