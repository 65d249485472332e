"""List kernels whose global loads the compiler serialised (load -> s_waitcnt vmcnt -> use, one memory round trip per element).
usage: python tools/scan_serial_loads.py [file.hip ...]   (default: every csrc/*.hip; compiles with --save-temps into /tmp/ppv_isa)
Output per kernel: number of global loads, number of "lone load then wait" pairs, and the head of its load / wait / store / branch
sequence (L = global load, w = s_waitcnt vmcnt, S = global store, | = branch).  A kernel with many `Lw` pairs in a row is a candidate
for the fix of DESIGN 4b: clamped addresses, all loads into distinct registers, __builtin_amdgcn_sched_barrier(0), then the masks."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "privacy-preserving-vision_amd", "csrc")
files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
out = "/tmp/ppv_isa"
os.makedirs(out, exist_ok=True)
for f in files:
    base = os.path.splitext(os.path.basename(f))[0]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, "--save-temps", "-c", f, "-o", os.path.join(out, base + ".o")], cwd=out, stderr=subprocess.DEVNULL, check=True)
    asm = os.path.join(out, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    L = open(asm).read().split("\n")
    starts = [(i, re.match(r"^(_ZN3ppv\w+):\s*;", l).group(1)) for i, l in enumerate(L) if re.match(r"^_ZN3ppv\w+:\s*;", l)]
    print("==", base)
    for start, name in starts:
        seq = []
        for l in L[start + 1:]:
            t = l.strip().split(";")[0].strip()
            if t.startswith(".Lfunc_end"):
                break
            if not t:
                continue
            k = t.split()[0]
            if k.startswith("global_load") and "lds" not in k:
                seq.append("L")
            elif k.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("w")
            elif k.startswith("global_store"):
                seq.append("S")
            elif k.startswith("s_cbranch"):
                seq.append("|")
        sq = "".join(seq)
        lone = len(re.findall(r"(?<!L)Lw", sq))
        if sq.count("L") >= 3 and lone >= 2:
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:70]
            print(f"  {demangled:70s} loads {sq.count('L'):3d}  lone {lone:3d}  {sq[:70]}")
