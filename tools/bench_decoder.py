"""Config-3 decoder alone at full size: B=128, 36x36x2048 encoder output, 512-d attention/LSTM/embedding, 9490 words."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ppv_amd.decoder as pd

B = int(os.environ.get("B", 128)); V = 9490; L = 52
torch.manual_seed(0)
dec = pd.DecoderWithAttention(512, 512, 512, V, encoder_dim=2048, dropout=0.5).cuda().train()
enc = torch.randn(B, 36, 36, 2048, device="cuda", requires_grad=True)
caps = torch.randint(0, V, (B, L), device="cuda")
# COCO-like caption lengths (<start> .. <end>): mean ~12.5 words
caplens = torch.randint(9, 19, (B, 1), device="cuda")
crit = torch.nn.CrossEntropyLoss()
from torch.nn.utils.rnn import pack_padded_sequence

def step():
    enc.grad = None
    for p in dec.parameters():
        p.grad = None
    scores, caps_sorted, dec_len, alphas, _ = dec(enc, caps, caplens)
    targets = caps_sorted[:, 1:]
    s = pack_padded_sequence(scores, dec_len, batch_first=True).data
    t = pack_padded_sequence(targets, dec_len, batch_first=True).data
    loss = crit(s, t) + ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    loss.backward()
    return sum(dec_len)

for _ in range(2):
    n = step()
torch.cuda.synchronize()
t0 = time.time()
K = 5
for _ in range(K):
    n = step()
torch.cuda.synchronize()
ms = (time.time() - t0) / K * 1e3
print(f"decoder fwd+bwd B={B}: {ms:.2f} ms/step, {n} decode positions, {B / ms * 1e3:.0f} img/s")
