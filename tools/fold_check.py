import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppv_amd
from ppv_amd.encoder import Encoder
torch.manual_seed(0)
enc = Encoder().cuda().train()
for B, H in ((4, 64), (16, 128)):
    img = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(1)).cuda()
    res = {}
    for mode in ("0", "0:b", "1:1", "1:2", "1:4"):
        os.environ["PPV_BN_FOLD_ACT"] = mode[0]
        if mode[0] == "1": os.environ["PPV_BN_FOLD_ROWS"] = mode[2]
        sd = {k: v.clone() for k, v in enc.state_dict().items()}
        out = enc(img.requires_grad_(True))
        fn = out.grad_fn
        res[mode] = [(sv[2].clone(), sv[5].clone(), sv[8].clone(), sv[11].float().clone()) for sv in fn.blocks]
        enc.load_state_dict(sd)
    for mode in ("0:b", "1:1", "1:2", "1:4"):
        worst = 0; wy = 0
        for a, b in zip(res["0"], res[mode]):
            for i in range(3):
                worst = max(worst, float((a[i][:2] - b[i][:2]).abs().max() / a[i][:2].abs().max()))
            wy = max(wy, float((a[3] - b[3]).norm() / a[3].norm()))
        first = next((i for i, (a, b) in enumerate(zip(res["0"], res[mode])) if float((a[3] - b[3]).norm() / a[3].norm()) > 1e-3), None)
        print(B, H, mode, "coef max rel diff %.3e   yout rel-L2 %.3e  first block off: %s" % (worst, wy, first))
