"""Where does the fp32 product mode's backward deviate from the oracle?  Per block: forward tap error and gradient-at-tap error ([1,1,1,1] trunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppv_amd.encoder import Encoder
from oracle.resnet import Encoder as OEncoder, Bottleneck as OBottleneck

layers, B, hw = (1, 1, 1, 1), 4, 64
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(SEED)
enc = Encoder(layers=layers, precision="fp32").cuda().train()
ref = OEncoder(round_bf16=False, layers=layers)
ref.load_state_dict({k: v.detach().cpu() for k, v in enc.state_dict().items()}, strict=True)
ref.train()
for p in list(enc.parameters()) + list(ref.parameters()):
    p.requires_grad_(True)
img = torch.rand(B, 3, hw, hw, generator=torch.Generator().manual_seed(100 + SEED))
w = torch.randn(B, 36, 36, 2048, generator=torch.Generator().manual_seed(5))
otaps = []
def hook(m, i, o):
    o.retain_grad(); otaps.append(o)
hs = [ref.resnet[3].register_forward_hook(hook)] + [m.register_forward_hook(hook) for m in ref.modules() if isinstance(m, OBottleneck)]
xi = img.clone().requires_grad_(True)
(ref(xi) * w).sum().backward()
taps = []
xg = img.cuda().requires_grad_(True)
out = enc._forward_fp32(xg, taps=taps)
for t in taps:
    t.retain_grad()
(out * w.cuda()).sum().backward()
rl2 = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
for i, (a, b) in enumerate(zip(taps, otaps)):
    print(f"tap {i}: fwd rel L2 {rl2(a.detach(), b.detach().permute(0, 2, 3, 1)):.1e}   grad rel L2 {rl2(a.grad, b.grad.permute(0, 2, 3, 1)):.1e}  shape {tuple(a.shape)}")
print(f"input grad rel L2 {rl2(xg.grad, xi.grad):.1e}")
rp = dict(ref.named_parameters())
for n, p in enc.named_parameters():
    e = rl2(p.grad, rp[n].grad)
    if e > 5e-4:
        print(f"   {n:40s} {e:.1e}   |ref| {rp[n].grad.norm().item():.3e}  |ref|_max {rp[n].grad.abs().max().item():.3e}")
# gradient at the OUTPUT of block 6.0's bn1 / conv1 through hooks on the oracle and a replay on the product side

