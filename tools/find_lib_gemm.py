"""Which torch op of a bench step reaches a BLAS library?  Runs one step (default: config 3, --decoder) under a TorchDispatchMode that
prints every aten matrix-product op (mm / addmm / bmm / baddbmm / mv / addmv / dot / linear / matmul) with the Python stack that issued it.
usage: python tools/find_lib_gemm.py [--headline]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

NAMES = ("mm", "addmm", "bmm", "baddbmm", "mv", "addmv", "dot", "linear", "matmul", "addbmm", "_scaled_mm", "einsum", "tensordot")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name in NAMES:
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            print(f"[lib-gemm] aten.{name} {shapes}", flush=True)
            for fr in traceback.extract_stack()[-9:-1]:
                if "site-packages/torch" not in fr.filename and "dist-packages/torch" not in fr.filename:
                    print(f"      {os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}", flush=True)
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(1234)
camera, encoder = bench.build(dev, global_max_sync=False)
decoder = None
if "--headline" not in sys.argv:
    from ppv_amd.decoder import DecoderWithAttention
    torch.manual_seed(3)
    decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(dev).train()
step, _ = bench.make_step(camera, encoder, 128, dev, None, decoder, False, graph=False)
step(); step()
torch.cuda.synchronize()
print("---- spying on one step", flush=True)
with Spy():
    step()
torch.cuda.synchronize()
print("---- done", flush=True)
