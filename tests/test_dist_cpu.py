"""CPU, world_size 2, gloo: the data-parallel gradient averaging layer (ppv_amd.dist_sync.GradSync) -- bucketing,
push order, flush -- gives every rank the mean of the per-rank gradients (SURVEY 8e oracle: N-rank averaged grads ==
1-rank grads on the concatenated batch, for a loss that is a mean over samples)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ppv_amd  # noqa: F401
    from ppv_amd.dist_sync import GradSync
    torch.manual_seed(0)
    w = [torch.randn(300, 7), torch.randn(5000), torch.randn(64, 64, 3, 3), torch.randn(3)]
    data = torch.randn(8, 11, generator=torch.Generator().manual_seed(1))           # global batch of 8
    proj = [torch.randn(11, t.numel(), generator=torch.Generator().manual_seed(2 + i)) for i, t in enumerate(w)]

    def grads_for(rows):
        ps = [t.clone().requires_grad_(True) for t in w]
        loss = sum(((rows @ pr) * p.reshape(1, -1)).sum(1) for p, pr in zip(ps, proj)).mean()
        loss.backward()
        return [p.grad for p in ps]

    full = grads_for(data)
    local = grads_for(data[rank * 4:(rank + 1) * 4])
    sync = GradSync(bucket_mb=0.01)                  # tiny buckets: several launches
    for g in reversed(local):
        sync.push(g)
    sync.flush()
    ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(local, full)) and sync.launched >= 2
    # the pre-flattened path: gradients written into bucket slices, reduced in place when a bucket's last slice is marked
    ps = [torch.nn.Parameter(t.clone()) for t in w]
    sync2 = GradSync(bucket_mb=0.01)
    sync2.attach(list(reversed(ps)))
    local2 = grads_for(data[rank * 4:(rank + 1) * 4])
    for p, g in reversed(list(zip(ps, local2))):
        v = sync2.grad_view(p)
        v.copy_(g)
        p.grad = v
        sync2.mark_ready(p)
    sync2.end_of_backward()
    ok = ok and sync2.launched >= 2 and all(torch.allclose(p.grad, b, rtol=1e-5, atol=1e-6) for p, b in zip(ps, full))
    ok = ok and all(p.grad.data_ptr() == sync2.grad_view(p).data_ptr() for p in ps)
    m = torch.tensor([float(rank + 1)])
    dist.all_reduce(m, op=dist.ReduceOp.MAX)         # the scalar exchange of Lens.py:312 (global_max_sync)
    ret[rank] = bool(ok and m.item() == world)
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
