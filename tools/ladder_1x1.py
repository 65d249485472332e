"""VERDICT r5 task 3: the 1x1 forward class, rung by rung (profiles/r06_1x1_ladder.json).  Compiles and runs tools/micro/ladder_1x1.hip (the
memory-system ladder: pattern -> + kernel's grid / LDS / sharing -> + weights -> + epilogue -> + LDS reads) and times the REAL kernels of
csrc/conv_gemm.hip on the same two layer-3 shapes in their three modes (PPV_CONV_DEBUG: loads only / K loop only / whole launch, each in
its own process: the switch is read once), cold rotating operands, rocprofv3-free event timing of back-to-back launches.
GPU box: python tools/ladder_1x1.py [out.json]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch
import ppv_amd.convops as co
B, H, NB = 128, 16, 6
M = B * H * H
def timed(fn, n=24):
    for _ in range(NB): fn()
    torch.cuda.synchronize()
    meds = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        meds.append(e0.elapsed_time(e1) / n * 1e3)
    return round(sorted(meds)[1], 1)
out = {}
for cin, cout in ((1024, 256), (256, 1024)):
    xs = [torch.randn(B, H, H, cin, device="cuda").bfloat16() for _ in range(NB)]
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.03
    wf = co.weight_layout(w, 0)
    part = torch.zeros(co.stat_tiles(M), 2, cout, device="cuda")
    it = [0]
    def f():
        it[0] += 1
        return co.conv_fwd(xs[it[0] %% NB], wf, 1, 0, stat_part=part)
    out["%%d_to_%%d" %% (cin, cout)] = timed(f)
print(json.dumps(out))
''' % ROOT


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_1x1_ladder.json")
    exe = "/tmp/ppv_ladder_1x1"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-Wno-unused-value", os.path.join(ROOT, "tools", "micro", "ladder_1x1.hip"), "-o", exe])
    micro = json.loads(subprocess.run([exe], capture_output=True, text=True, check=True, timeout=300).stdout)
    real = {}
    for mode, name in ((0, "whole_launch"), (1, "loads_only"), (2, "k_loop_only")):
        env = dict(os.environ, PPV_CONV_DEBUG=str(mode))
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=300)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        real[name] = json.loads(rows[-1]) if rows else {"error": r.stderr[-400:]}
    doc = {"what": "1x1 forward class ladder, layer-3 shapes at B = 128 (M = 32768), us per launch, cold rotating operands (6 sets), median of 3 loops; "
                   "micro = tools/micro/ladder_1x1.hip (no MFMA), real = conv_gemm_pipe_kernel<256,128,3,64,1> (1024 -> 256) / <256,128,3,32,2> (256 -> 1024) "
                   "with the BatchNorm statistics epilogue, PPV_CONV_DEBUG modes",
           "micro": micro, "real_kernel_us": real}
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
