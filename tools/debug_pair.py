"""Debug aid for tests/test_wgrad_pair_gpu.py: the executor child under PPV_WGRAD_PAIR = 0 / 0 / 1, the tensors whose gradient norms differ."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_wgrad_pair_gpu import _CHILD
outs = []
for flag in ("0", "0", "1", "1"):
    r = subprocess.run([sys.executable, "-c", _CHILD], env=dict(os.environ, PPV_WGRAD_PAIR=flag), capture_output=True, text=True, timeout=600)
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not rows:
        print("flag", flag, "no rows", r.stderr[-1500:]); sys.exit(1)
    outs.append(json.loads(rows[-1]))
def cmp(a, b, tag):
    d = sorted(((abs(a[k][1] - b[k][1]) / (abs(a[k][1]) + 1e-12), k, a[k][1], b[k][1]) for k in a), reverse=True)
    print(tag, [(f"{x[0]:.2e}", x[1], f"{x[2]:.4e}", f"{x[3]:.4e}") for x in d[:6]])
cmp(outs[0], outs[1], "0 vs 0:")
cmp(outs[0], outs[2], "0 vs 1:")
cmp(outs[2], outs[3], "1 vs 1:")
