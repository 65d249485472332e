"""One training iteration of the reference's Image_Caption/train.py:259-323 on the MI355X modules, with synthetic data
(no dataset, no checkpoints): camera -> encoder -> decoder -> the reference's loss mix -> three Adam steps.

    python examples/caption_train_step.py            # needs an MI355X and the built library (__graft_entry__.build())

The lines marked `train.py:N` are the reference's; everything else is glue a maintainer already has."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn.utils.rnn import pack_padded_sequence

import ppv_amd  # noqa: F401
from ppv_amd.camera_lens import OpticsZernike          # from Camera.Lens import OpticsZernike      (train.py:11)
from ppv_amd.encoder import Encoder                    # from models import Encoder, DecoderWithAttention (train.py:12)
from ppv_amd.decoder import DecoderWithAttention

device = torch.device("cuda", 0)
vocab = 9490
camera = OpticsZernike(input_shape=[None, 256, 256, 3], device=device, zernike_terms=350, patch_size=256, height_tolerance=2e-8,
                       sensor_distance=0.025, wave_resolution=[896, 896], sample_interval=3e-06, upsample=False)   # train.py:64-66
encoder = Encoder().to(device)                                                                                    # train.py:103
decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=vocab, dropout=0.3).to(device)
encoder.fine_tune(True)
camera_optimizer = torch.optim.Adam([p for p in camera.parameters() if p.requires_grad], lr=5e-7)
encoder_optimizer = torch.optim.Adam([p for p in encoder.parameters() if p.requires_grad], lr=1e-4)
decoder_optimizer = torch.optim.Adam([p for p in decoder.parameters() if p.requires_grad], lr=4e-4)
criterion = torch.nn.CrossEntropyLoss().to(device)

B = 32
imgs = torch.rand(B, 3, 256, 256, device=device)
caps = torch.randint(0, vocab, (B, 52), device=device)
caplens_cpu = torch.randint(9, 19, (B, 1))          # what the data loader yields (datasets.py:60)
caplens = caplens_cpu.to(device)                    # train.py:263
camera.train(); encoder.train(); decoder.train()                                                                  # train.py:245-247

for it in range(3):
    # optional, ppv_amd only: the decoder gets the lengths from the loader's CPU tensor, so its forward never waits for the device
    # (models.py:193's .tolist() on a device tensor waits for the whole camera + encoder forward queued before it)
    decoder.stage_lengths(caplens, host=caplens_cpu)
    imgs_sensor, psf, coeffs, loss_psf = camera(imgs, None, "3")                                                  # train.py:270
    imgs_encoded = encoder(imgs_sensor)                                                                           # train.py:272
    scores, caps_sorted, decode_lengths, alphas, sort_ind = decoder(imgs_encoded, caps, caplens)                  # train.py:274
    targets = caps_sorted[:, 1:]
    scores_p = pack_padded_sequence(scores, decode_lengths, batch_first=True)                                     # train.py:277-278
    targets_p = pack_padded_sequence(targets, decode_lengths, batch_first=True)
    loss_ce = criterion(scores_p.data, targets_p.data)                                                            # train.py:280
    loss_dsr = 1.0 * ((1.0 - alphas.sum(dim=1)) ** 2).mean()                                                      # train.py:281
    loss_cam = 1 - torch.nn.functional.mse_loss(imgs, imgs_sensor)
    loss = 0.4 * (loss_ce + loss_dsr) + 6 * loss_cam + 30 * loss_psf
    for opt in (decoder_optimizer, encoder_optimizer, camera_optimizer):
        opt.zero_grad()
    loss.backward()
    camera_optimizer.step()
    for opt in (decoder_optimizer, encoder_optimizer):                                                            # clip_gradient, train.py:311-316
        for group in opt.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.grad.data.clamp_(-5.0, 5.0)
        opt.step()
    camera.zernike_coeffs_train[1:].data.clamp_(-1, 1)                                                            # train.py:322-323
    print(f"iter {it}: loss {loss.item():.4f}  ce {loss_ce.item():.4f}  psf {float(loss_psf.detach()):.4e}")
