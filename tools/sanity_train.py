"""Stability / race screen: 120 optimiser steps of the bench step (all stream overlaps on) with a fixed batch; the loss must
stay finite and go down, and a re-run with every overlap off must reproduce the first steps' losses to bf16 noise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


WITH_DECODER = "--decoder" in sys.argv


def run(steps, batch=32):
    torch.manual_seed(1234)
    dev = torch.device("cuda", 0)
    camera, encoder = bench.build(dev, global_max_sync=False)
    decoder = None
    if WITH_DECODER:
        from ppv_amd.decoder import DecoderWithAttention
        torch.manual_seed(3)
        decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.0).to(dev).train()
    step, _ = bench.make_step(camera, encoder, batch, dev, None, decoder)
    out = []
    for i in range(steps):
        out.append(float(step().detach()))
    torch.cuda.synchronize()
    return out


a = run(120)
print("overlap on :", [round(v, 4) for v in a[:3]], "...", [round(v, 4) for v in a[-3:]])
assert all(map(lambda v: v == v and abs(v) < 1e6, a)), "non-finite loss"
assert a[-1] < a[0], "loss did not go down"
os.environ["PPV_WGRAD_SIDE"] = "0"
os.environ["PPV_OPT_OVERLAP"] = "0"
b = run(4)
print("overlap off:", [round(v, 4) for v in b])
assert max(abs(x - y) for x, y in zip(a[:4], b)) < 2e-2 * max(1.0, abs(a[0])), "overlapped and serial runs diverge early"
print("ok")
