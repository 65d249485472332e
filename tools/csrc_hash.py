"""sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h: names + contents, sorted).  tools/collect_pmc.py and
collect_mfma.py store it in the profile JSON; bench.py prints counter-derived fields (`roofline.traffic`, `mfma_busy_frac_pmc`,
`step_hbm`) only when the stored hash equals the hash of the sources the running library was built from -- a kernel change without
a new `tools/profile_round.sh` pass yields null + "stale": true instead of a stale number.  (The GPU box has no .git, so this is a
content hash, not `git rev-parse`.)"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "privacy-preserving-vision_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "privacy-preserving-vision_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
