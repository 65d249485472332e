"""Host-side logic of the Encoder drop-in and of GradSync that needs no GPU (r2 advisor findings)."""
import io

import torch


def test_cached_parameter_list_follows_replaced_parameters():
    import ppv_amd  # noqa: F401
    from ppv_amd.encoder import Encoder
    e = Encoder(3, layers=(1, 1, 1, 1))
    l1 = e._param_list()
    assert e._param_list() is l1                                    # cached

    class Parent(torch.nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.enc = enc

    p = Parent(e)
    p.load_state_dict({k: v.clone() for k, v in p.state_dict().items()}, assign=True)     # never calls Encoder.load_state_dict
    l2 = e._param_list()
    assert l2 is not l1 and all(a is b for a, b in zip(l2, e.resnet.parameters()))
    e.resnet[5][0].conv1.weight = torch.nn.Parameter(e.resnet[5][0].conv1.weight.detach().clone())   # weight surgery
    l3 = e._param_list()
    assert l3 is not l2 and all(a is b for a, b in zip(l3, e.resnet.parameters()))
    buf = io.BytesIO()
    torch.save(e, buf)                                              # whole-module pickle (train.py:145) still works
    buf.seek(0)
    e2 = torch.load(buf, weights_only=False)
    assert [n for n, _ in e2.named_parameters()] == [n for n, _ in e.named_parameters()]


def test_gradsync_attach_resets_bucket_counters_and_accepts_no_parameters():
    import torch.distributed as dist
    import ppv_amd  # noqa: F401
    from ppv_amd.dist_sync import GradSync
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    try:
        ps = [torch.nn.Parameter(torch.randn(10)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(3))]
        s = GradSync(bucket_mb=1)
        s.attach(ps)
        s.grad_view(ps[0]).fill_(1.0)
        s.mark_ready(ps[0])                                         # a backward that aborts after one of three slices ...
        assert s._pending == [2] and s.launched == 0
        s.attach(ps)                                                # ... and the next step starts
        assert s._pending == [3]
        for p in ps:
            s.grad_view(p).fill_(2.0)
            s.mark_ready(p)
        assert s.launched == 1 and s._pending == [3]                # fired exactly when ALL slices were written
        s.attach([])                                                # fine_tune(False): nothing to own
        assert s._flat == [] and s.grad_view(ps[0]) is None
        s.end_of_backward()
    finally:
        dist.destroy_process_group()
