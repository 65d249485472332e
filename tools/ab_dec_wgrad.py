"""In-process A/B of the decoder's batched weight-gradient paths through the config-3 step (one box, one process, interleaved rounds:
cdna_hip_programming.md 5.4 rule 24).  usage: python tools/ab_dec_wgrad.py [modes...]   (default: lib x3 hip)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

modes = sys.argv[1:] or ["lib", "x3", "hip"]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
from ppv_amd.decoder import DecoderWithAttention
torch.manual_seed(3)
decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(dev).train()
step, _ = bench.make_step(camera, encoder, 128, dev, None, decoder, False, graph=False)
for m in modes:
    os.environ["PPV_DEC_WGRAD"] = m
    for _ in range(3):
        step()
torch.cuda.synchronize()
res = {m: [] for m in modes}
for rnd in range(4):
    for m in modes:
        os.environ["PPV_DEC_WGRAD"] = m
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        res[m].append((time.perf_counter() - t0) / 10 * 1e3)
for m in modes:
    v = sorted(res[m])
    print(f"PPV_DEC_WGRAD={m:5s} ms/step per round {[round(x, 3) for x in res[m]]}  median {v[len(v) // 2]:.3f}  -> {128 / v[len(v) // 2] * 1e3:.1f} images/s", flush=True)
