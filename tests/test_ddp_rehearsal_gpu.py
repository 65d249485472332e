"""Two ranks on the one visible GPU (gloo transport, rendezvous on 127.0.0.1) through the real bench step: after every step
all ranks must hold bit-identical gradients for every encoder / lens parameter (bench.py PPV_CHECK_SYNC) -- this is the check
that catches a gradient bucket that was never all-reduced or an optimiser step that did not wait for the side streams."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra,port", [((), "29541"), (("--decoder",), "29542")])
def test_two_rank_step_keeps_gradients_identical_across_ranks(extra, port):
    env = dict(os.environ, PPV_DIST_BACKEND="gloo", PPV_FORCE_DEVICE0="1", PPV_CHECK_SYNC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "2", "--warmup", "1",
           "--no-roofline", *extra]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["value"] > 0


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts the two ranks itself (child torchrun) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PPV_DIST_BACKEND="gloo", PPV_FORCE_DEVICE0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--steps", "2", "--warmup", "1",
                          "--no-roofline"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
