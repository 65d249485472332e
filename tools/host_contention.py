"""8-GPU host rehearsal on a box with ONE GPU (r3 verdict 8b): N processes, each pinned to its own CORES_PER_RANK cores, build the bench
step and enqueue it at the same moment -- what N interpreters on one host cost each other (memory bandwidth, the HIP runtime's
threads, the kernel driver's submission path).  Every rank measures the host time to enqueue one step into an EMPTY queue (2 steps
after a synchronise, repeated), the number that must stay below the device's ~22 ms for the device to set the pace.

    python tools/host_contention.py [N=6] [cores_per_rank=2] [batch=8] [hogs=0]      # parent: spawns the ranks, prints one JSON line

hogs (round 6): that many extra processes, each pinned to the next cores_per_rank cores, run an interpreter-bound loop for the duration of
the "together" round WITHOUT touching the GPU -- the host load of ranks 7 and 8 of an 8-GPU node on a box whose card admits six processes.

The device is shared here, so a small batch keeps the device time of a step short (the launch sequence, and with it the host work,
is the same at any batch); at most 6 processes may use the card on the GPU box."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(rank, n, cores, batch, go_at):
    first = rank * cores
    try:
        os.sched_setaffinity(0, set(range(first, first + cores)))
    except Exception as e:                                   # fewer cores than asked for: unpinned, say so
        print(f"[rank {rank}] affinity not set: {e}", file=sys.stderr)
    sys.path.insert(0, ROOT)
    import torch
    import bench
    torch.set_num_threads(cores)
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    camera, encoder = bench.build(dev, global_max_sync=False)
    step, _ = bench.make_step(camera, encoder, batch, dev, None)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    while time.time() < go_at:                               # all ranks start their timed rounds together
        time.sleep(0.005)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(); step()
        ts.append((time.perf_counter() - t0) / 2 * 1e3)
    torch.cuda.synchronize()
    ts.sort()
    print(json.dumps({"rank": rank, "enqueue_ms_median": round(ts[len(ts) // 2], 3), "enqueue_ms_min": round(ts[0], 3), "enqueue_ms_max": round(ts[-1], 3)}),
          flush=True)


def hog(rank, cores, until):
    first = rank * cores
    try:
        os.sched_setaffinity(0, set(range(first, first + cores)))
    except Exception:
        pass
    d, n = {}, 0
    while time.time() < until:                                # interpreter + allocator traffic, like a rank's enqueue loop
        for i in range(20000):
            d[i & 1023] = [i, str(i)]
        n += 1
    print(json.dumps({"hog": rank, "loops": n}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(*(int(v) for v in sys.argv[2:6]), float(sys.argv[6]))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--hog":
        hog(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cores = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    hogs = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out = {"ranks": n, "cores_per_rank": cores, "batch": batch, "host_cores": os.cpu_count(), "cpu_only_hogs": hogs}
    for label, count in (("alone", 1), ("together", n)):
        go_at = time.time() + (75 if count > 1 else 60)      # model build + first torch import on a fresh box
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), str(n), str(cores), str(batch), str(go_at)],
                                  stdout=subprocess.PIPE, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(count)]
        hps = []
        if count > 1:
            hps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--hog", str(n + h), str(cores), str(go_at + 30)], stdout=subprocess.PIPE, text=True)
                   for h in range(hogs)]
        res = []
        for p in hps:
            p.communicate(timeout=600)
        for p in procs:
            o, _ = p.communicate(timeout=600)
            res += [json.loads(l) for l in o.splitlines() if l.startswith("{")]
        out[label] = sorted(res, key=lambda d: d["rank"])
        print(f"[host_contention] {label}: " + ", ".join(f"{d['enqueue_ms_median']:.2f}" for d in out[label]), file=sys.stderr, flush=True)
    med = lambda rs: sorted(d["enqueue_ms_median"] for d in rs)[len(rs) // 2]
    out["alone_ms"] = med(out["alone"])
    out["together_ms_median"] = med(out["together"])
    out["together_ms_worst"] = max(d["enqueue_ms_median"] for d in out["together"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
