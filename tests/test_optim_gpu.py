"""ppv_amd.optim.Adam against torch.optim.Adam on the CPU (the oracle here is stock PyTorch's single-tensor f32 implementation of the
optimiser the reference's harness constructs, Image_Caption/train.py:92-101): same parameters, same gradients, five steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(seed, sizes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in sizes]


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_adam_equals_torch_cpu(wd):
    from ppv_amd.optim import Adam
    sizes = [(64,), (64, 3, 7, 7), (256, 64, 1, 1), (5,), (4099,), (512, 512, 3, 3), (3, 1367)]
    ref = [torch.nn.Parameter(t.clone()) for t in _make(0, sizes)]
    flat = torch.zeros(sum(p.numel() for p in ref) + 3, device="cuda")           # gradients as slices of one buffer, some of them unaligned
    got, off = [], 1
    for t in _make(0, sizes):
        got.append(torch.nn.Parameter(t.cuda()))
    o_ref = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=False)
    o_got = Adam(got, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for step in range(5):
        grads = _make(100 + step, sizes)
        off = 1
        for p, q, g in zip(ref, got, grads):
            p.grad = g.clone()
            view = flat[off:off + g.numel()].view_as(g)
            view.copy_(g)
            q.grad = view
            off += g.numel()
        if step == 3:                                                            # a parameter without a gradient in one step keeps its count
            ref[3].grad = None
            got[3].grad = None
        o_ref.step()
        o_got.step()
    torch.cuda.synchronize()
    for p, q in zip(ref, got):
        assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=1e-7)
        assert torch.allclose(o_got.state[q]["exp_avg"].cpu(), o_ref.state[p]["exp_avg"], rtol=2e-6, atol=2e-7)     # one rounding of g - m at O(1)
        assert torch.allclose(o_got.state[q]["exp_avg_sq"].cpu(), o_ref.state[p]["exp_avg_sq"], rtol=2e-6, atol=1e-8)
        assert float(o_got.state[q]["step"]) == float(o_ref.state[p]["step"])


def test_adam_state_dict_round_trip_with_torch():
    from ppv_amd.optim import Adam
    ps = [torch.nn.Parameter(t.cuda()) for t in _make(1, [(300,), (64, 64)])]
    o = Adam(ps, lr=1e-3)
    for p in ps:
        p.grad = torch.randn_like(p)
    o.step()
    t = torch.optim.Adam(ps, lr=1e-3)
    t.load_state_dict(o.state_dict())                                            # a checkpoint written with one loads into the other
    o2 = Adam(ps, lr=1e-3)
    o2.load_state_dict(t.state_dict())
    assert float(o2.state[ps[0]]["step"]) == 1.0
    assert torch.equal(o2.state[ps[1]]["exp_avg"], o.state[ps[1]]["exp_avg"])


def test_adam_refuses_what_it_does_not_implement():
    from ppv_amd.optim import Adam
    p = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]
    with pytest.raises(NotImplementedError):
        Adam(p, amsgrad=True)
    p64 = [torch.nn.Parameter(torch.zeros(4, device="cuda", dtype=torch.float64))]
    o = Adam(p64)
    p64[0].grad = torch.zeros(4, device="cuda", dtype=torch.float64)
    with pytest.raises(NotImplementedError):
        o.step()


def test_adam_follows_a_reloaded_state_and_does_not_grow_its_tables():
    """the descriptor table holds the moment pointers too -- after load_state_dict() on an optimiser that has already stepped
    the loaded moments (new tensors) must be the ones that advance; parameters with different step counts reuse their table slots."""
    from ppv_amd.optim import Adam
    sizes = [(300,), (64, 16, 3, 3), (4099,)]
    ref = [torch.nn.Parameter(t.clone()) for t in _make(0, sizes)]
    got = [torch.nn.Parameter(t.cuda()) for t in _make(0, sizes)]
    o_ref = torch.optim.Adam(ref, lr=1e-2, foreach=False)
    o_got = Adam(got, lr=1e-2)

    def both(step, skip=None):
        for i, (p, q, g) in enumerate(zip(ref, got, _make(200 + step, sizes))):
            p.grad, q.grad = (None, None) if i == skip else (g.clone(), g.cuda())
        o_ref.step()
        o_got.step()

    both(0)
    both(1)
    snap_ref, snap_got = {k: v for k, v in o_ref.state_dict().items()}, o_got.state_dict()
    import copy
    snap_ref, snap_got = copy.deepcopy(snap_ref), copy.deepcopy(snap_got)
    w_ref, w_got = [p.detach().clone() for p in ref], [q.detach().clone() for q in got]
    both(2)
    both(3)
    # roll back: parameters and optimiser state of step 2, then step again with the same gradients
    with torch.no_grad():
        for p, w in zip(ref, w_ref):
            p.copy_(w)
        for q, w in zip(got, w_got):
            q.copy_(w)
    o_ref.load_state_dict(snap_ref)
    o_got.load_state_dict(snap_got)
    both(2)
    both(3, skip=1)                                                              # split step counts from here on
    both(4)
    n_tables = len(o_got._tables)
    for s in range(5, 12):
        both(s)
    torch.cuda.synchronize()
    for p, q in zip(ref, got):
        assert torch.allclose(p.detach(), q.detach().cpu(), rtol=2e-5, atol=2e-6)
        a, b = o_ref.state[p], o_got.state[q]
        assert float(a["step"]) == float(b["step"])
        assert torch.allclose(a["exp_avg"], b["exp_avg"].cpu(), rtol=2e-5, atol=1e-7)
        assert torch.allclose(a["exp_avg_sq"], b["exp_avg_sq"].cpu(), rtol=2e-5, atol=1e-9)
    assert len(o_got._tables) == n_tables <= 3                                   # (group, slot) keys are reused: no growth with the step count
