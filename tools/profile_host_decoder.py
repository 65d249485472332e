"""Host enqueue profile of the config-3 step (camera + trunk + attention decoder): cProfile of 3 steps, autograd on the calling thread."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ppv_amd.decoder import DecoderWithAttention
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
torch.manual_seed(3)
decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(dev)
decoder.train()
step, _ = bench.make_step(camera, encoder, 128, dev, None, decoder)
for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
