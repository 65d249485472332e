#!/usr/bin/env python3
"""Benchmark of the hot path: images/sec, forward + backward (+ optimiser), learned-optics Camera + ResNet-101
encoder at 256 x 256 on N MI355X (BASELINE.json metric).  One process per GPU; RCCL gradient all-reduce for N > 1.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step reproduces reference Image_Caption/train.py:259-323 without the caption decoder (a "next" row):
camera(imgs, None, "3") -> encoder(sensor) -> loss = 0.4 * head(enc_out) + 6 * (1 - MSE(imgs, sensor)) + 30 * loss_psf
-> zero_grad -> backward -> camera Adam(5e-7) -> clamp encoder grads to +-5 -> encoder Adam(1e-4) -> clamp coeffs.
Rank 0 prints ONE JSON line (contract in the task statement), with `roofline` (the dominant MFMA conv kernel, timed
live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0, N = 1).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0          # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
TRUNK_GFLOP_PER_IMG = 59.07              # SURVEY 8d / BASELINE.md: fwd 20.37 + dgrad 20.37 + wgrad 18.32 (layer2-4)


def build(device, global_max_sync):
    import ppv_amd  # noqa: F401
    from ppv_amd.camera_lens import OpticsZernike
    from ppv_amd.encoder import Encoder
    camera = OpticsZernike(input_shape=[None, 256, 256, 3], device=device, zernike_terms=350, patch_size=256,
                           height_tolerance=2e-8, sensor_distance=0.025, wave_resolution=[896, 896],
                           sample_interval=3e-06, upsample=False, coeff_layout="B", global_max_sync=global_max_sync)
    gold = os.path.join(ROOT, "tests", "golden", "ic_real.npz")
    if os.path.exists(gold):                                   # the coefficients of the reference's Camera/Model.pth
        c = torch.tensor(np.load(gold)["modelpth_coeffs"]).reshape(-1, 1, 1)
        camera.load_state_dict({"zernike_coeffs_no_train": c[:3], "zernike_coeffs_train": c[3:]})
    torch.manual_seed(2)
    encoder = Encoder().to(device)
    encoder.train()
    camera.train()
    return camera, encoder


class _MeanSquare(torch.autograd.Function):
    """mean(x^2) on the dense [B,E,E,2048] tensor with a one-pass forward (read) and a one-pass backward (read + write): what a
    foreign consumer of encoder_out costs (PPV_BENCH_DENSE_HEAD=1; 1.36 GB each way at B = 128)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.linalg.vector_norm(x) ** 2 / x.numel()

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return x * (g * (2.0 / x.numel()))


_POOL_GRAM = {}


def _pool_gram(h, e, device):
    """P^T P for AdaptiveAvgPool1d(h -> e) as an [e, h] matrix P (models.py:27: windows floor(i h / e) .. ceil((i + 1) h / e));
    built once per geometry (a host -> device copy inside the step would synchronise it)."""
    key = (h, e, str(device))
    if key not in _POOL_GRAM:
        p = torch.zeros(e, h, dtype=torch.float32)
        for i in range(e):
            lo, hi = (i * h) // e, -((-(i + 1) * h) // e)
            p[i, lo:hi] = 1.0 / (hi - lo)
        _POOL_GRAM[key] = (p.t() @ p).to(device)
    return _POOL_GRAM[key]


class _CellsMeanSquare(torch.autograd.Function):
    """The same mean(encoder_out^2), evaluated the way ppv_amd's own decoder consumes the encoder (decoder.py "compact path"):
    on the 8x8 map behind the up-sampled output.  encoder_out = (P (x) P) cells per channel, so
    sum(encoder_out^2) = <cells, (P^T P (x) P^T P) cells> and the gradient 2 (P^T P (x) P^T P) cells / n goes straight to the
    map: the 1.36 GB f32 tensor is written once by the encoder (its models.py:39-41 surface) and never read back."""

    @staticmethod
    def forward(ctx, cells, e):
        _, h, w, _ = cells.shape
        key = ("kron", h, w, e, str(cells.device))
        if key not in _POOL_GRAM:                                 # (P^T P) (x) (P^T P) as ONE [hw, hw] matrix: a single batched GEMM of
            _POOL_GRAM[key] = torch.kron(_pool_gram(h, e, cells.device), _pool_gram(w, e, cells.device)).contiguous()   # sane shape
        x = cells.float()
        t = torch.matmul(_POOL_GRAM[key], x.view(x.shape[0], h * w, x.shape[3])).view_as(x)
        n = cells.shape[0] * e * e * cells.shape[3]
        ctx.save_for_backward(t)
        ctx.n = n
        return (x * t).sum() / n

    @staticmethod
    def backward(ctx, g):
        t, = ctx.saved_tensors
        return t * (g * (2.0 / ctx.n)), None


_DENSE_HEAD = [False]     # True: the stand-in head consumes the dense tensor even when the encoder also hands over its cells


def head_stand_in(enc_out):
    """Stand-in for the caption head (CE + attention regulariser, train.py:276-282) in the headline metric."""
    cells = getattr(enc_out, "_ppv_cells", None)
    if cells is None or _DENSE_HEAD[0] or os.environ.get("PPV_BENCH_DENSE_HEAD"):
        return _MeanSquare.apply(enc_out)
    return _CellsMeanSquare.apply(cells, enc_out.shape[1])


@contextlib.contextmanager
def surface_mode(encoder, dense):
    """dense=True: the module surface of models.py:39-41 -- Encoder.forward materialises its [B,E,E,2048] f32 output, the head reads it and
    returns a dense gradient that the encoder pools back.  dense=False: ppv_amd's own consumer (the decoder's compact path): the dense
    tensor is lazy and the head works on the cells behind it."""
    lazy0 = getattr(encoder, "lazy_output", None)
    head0 = _DENSE_HEAD[0]
    if lazy0 is not None:
        encoder.lazy_output = not dense
    _DENSE_HEAD[0] = dense
    try:
        yield
    finally:
        if lazy0 is not None:
            encoder.lazy_output = lazy0
        _DENSE_HEAD[0] = head0


def timed_windows(step, k, n_windows, world, device, sync=None):
    """n_windows windows of EXACTLY k steps, each bracketed by a barrier + synchronize on both sides; per window the MAX over ranks.
    -> (seconds per window, host seconds spent enqueueing each window)."""
    sync = sync or torch.cuda.synchronize
    wins, enq = [], []
    for _ in range(n_windows):
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        e = time.perf_counter() - t0
        sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        wins.append(el)
        enq.append(e)
    return wins, enq


def surface_fields(world, batch, steps, wins, other_wins, headline_dense, has_decoder):
    """The timing-protocol keys of the JSON line: `value` is computed from the median of `wins`; this adds the windows, the protocol,
    and both output surfaces (value_dense_surface == value unless --lazy / --decoder)."""
    med = lambda w: None if not w else sorted(w)[len(w) // 2]
    head, other = med(wins), med(other_wins)
    dense, lazy = (head, other) if headline_dense else (other, head)
    if has_decoder:
        dense = lazy = None
    rate = lambda t: None if t is None else round(world * batch * steps / t, 1)
    ms = lambda t: None if t is None else round(t / steps * 1e3, 3)
    return {
        "windows_ms_per_step": [ms(w) for w in wins],
        "timing": f"median of {len(wins)} windows of exactly {steps} steps, each bracketed by barrier + synchronize; max over ranks per window",
        "gc": "cyclic GC frozen + disabled during warm-up and timed windows (gc.freeze(); gc.disable()), re-enabled after",
        "surface": ("dense: Encoder.forward writes the [B,36,36,2048] f32 tensor of models.py:39-41, the stand-in head reads it and returns a dense "
                    "gradient" if headline_dense else ("decoder on the encoder's cells (compact path)" if has_decoder else
                                                       "lazy: dense output not materialised, head on the 8x8 cells (--lazy)")),
        "value_dense_surface": rate(dense), "ms_per_step_dense_surface": ms(dense),
        "value_lazy_consumer": rate(lazy), "ms_per_step_lazy_consumer": ms(lazy),
        "windows_ms_per_step_other_surface": None if not other_wins else [ms(w) for w in other_wins],
    }


def make_step(camera, encoder, batch, device, sync, decoder=None, ssim_loss=False, graph=False):
    """graph=True: the step is going to be captured into a hipGraph (graph_step below): Adam keeps its step counters on the device
    (capturable) and the cross-STEP overlap of the encoder's Adam with the next camera forward is off (a captured step is
    self-contained)."""
    enc_params = [p for p in encoder.parameters() if p.requires_grad]
    cam_params = [p for p in camera.parameters() if p.requires_grad]
    # encoder / decoder optimisers (train.py:92-101): torch's fused Adam, as a user of the reference would construct it.  PPV_BENCH_PPV_ADAM=1:
    # ppv_amd.optim.Adam (same arithmetic and state layout, the whole list in one 10 k-workgroup launch: 193 us against 629 us alone on the
    # device) -- measured -0.04 ms on the headline and +0.45 ms with the decoder: the update runs BESIDE the next step's camera forward, where
    # torch's 45-240-workgroup launches leave the CUs to the critical path and a full-width launch does not
    if graph or os.environ.get("PPV_BENCH_PPV_ADAM", "0") == "0":
        def make_adam(ps, lr):
            return torch.optim.Adam(ps, lr=lr, fused=True, capturable=graph)
    else:
        from ppv_amd.optim import Adam as _PpvAdam

        def make_adam(ps, lr):
            return _PpvAdam(ps, lr=lr)
    opt_enc = make_adam(enc_params, 1e-4)
    opt_cam = torch.optim.Adam(cam_params, lr=5e-7, fused=os.environ.get("PPV_BENCH_CAM_ADAM_FUSED", "1") != "0", capturable=graph)
    rank = dist.get_rank() if dist.is_initialized() else 0
    imgs = torch.rand(batch, 3, 256, 256, generator=torch.Generator().manual_seed(rank), dtype=torch.float32).to(device)
    # PPV_BENCH_H2D=1: the PCIe-inclusive variant (never `value`): the batch starts in pinned host memory every step
    imgs_host = imgs.cpu().pin_memory() if os.environ.get("PPV_BENCH_H2D") else None
    if decoder is not None:                                                       # BASELINE.json config 3 / 5
        from torch.nn.utils.rnn import pack_padded_sequence
        dec_params = [p for p in decoder.parameters() if p.requires_grad]
        opt_dec = make_adam(dec_params, 4e-4)                                       # train.py:33,100-101
        gen = torch.Generator().manual_seed(100 + rank)
        caps = torch.randint(0, decoder.vocab_size, (batch, 52), generator=gen).to(device)
        caplens_host = torch.randint(9, 19, (batch, 1), generator=gen)            # COCO-like lengths incl. <start>/<end>: what the loader yields
        caplens = caplens_host.to(device)                                         # train.py:263

    opt_stream = torch.cuda.Stream(device=device) if (os.environ.get("PPV_OPT_OVERLAP", "1") != "0" and not graph) else None
    fused_mse = not ssim_loss and os.environ.get("PPV_BENCH_TORCH_MSE", "0") == "0"      # PPV_BENCH_TORCH_MSE=1: the three torch ops (A/B)
    if fused_mse:
        from ppv_amd.losses import camera_mse_tap

    def step():
        nonlocal imgs
        if imgs_host is not None:
            imgs = imgs_host.to(device, non_blocking=True)
        if decoder is not None and os.environ.get("PPV_DEC_STAGE_LENGTHS", "1") != "0":
            decoder.stage_lengths(caplens, host=caplens_host)   # the loader's CPU copy: forward() never fetches lengths from the device (decoder.py)
        sensor, psf, coeffs, loss_psf = camera(imgs, None, "3")
        if opt_stream is not None:             # the encoder's (and decoder's) Adam of the previous step ran beside the camera forward
            torch.cuda.current_stream().wait_stream(opt_stream)
        loss_cam = None
        if fused_mse:                          # train.py:284-288 on ppv_amd.losses: the encoder reads the sensor THROUGH the loss node, whose
            sensor, loss_cam = camera_mse_tap(imgs, sensor)     # backward adds the encoder's gradient to its own in one pass
        enc_out = encoder(sensor)
        if decoder is not None:                                                   # train.py:274-282
            scores, caps_sorted, dec_len, alphas, _ = decoder(enc_out, caps, caplens)
            sc = pack_padded_sequence(scores, dec_len, batch_first=True).data
            tg = pack_padded_sequence(caps_sorted[:, 1:], dec_len, batch_first=True).data
            loss_head = torch.nn.functional.cross_entropy(sc, tg) + 1.0 * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
            opt_dec.zero_grad(set_to_none=True)
        else:
            # stand-in for CE + attention regulariser: one read of encoder_out forward, one dense gradient backward
            loss_head = head_stand_in(enc_out)
        if ssim_loss:                                                             # camera_loss = 'SSIM', train.py:172-173
            from ppv_amd.ssim import ssim
            loss_cam = 1 - ssim(imgs, sensor)
        elif loss_cam is None:
            loss_cam = 1 - torch.nn.functional.mse_loss(imgs, sensor)
        loss = 0.4 * loss_head + 6 * loss_cam + 30 * loss_psf
        opt_enc.zero_grad(set_to_none=True)
        opt_cam.zero_grad(set_to_none=True)
        loss.backward()                        # encoder gradients are all-reduced inside backward (side stream)
        if sync is not None:
            sync.flush()                       # tail bucket + make this stream wait for every all-reduce
            sync.reduce_now([p.grad for p in cam_params])
            if os.environ.get("PPV_CHECK_SYNC"):                                  # rehearsal check: ranks agree on every gradient
                for p in enc_params + cam_params:
                    lo, hi = p.grad.detach().clone(), p.grad.detach().clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                    assert torch.equal(lo, hi), "gradient differs across ranks after the all-reduce"
        opt_cam.step()
        camera.zernike_coeffs_train[1:].data.clamp_(-1, 1)                        # train.py:322-323
        if decoder is not None and sync is not None:
            sync.reduce_now([p.grad for p in dec_params])
        # The encoder / decoder updates touch nothing the next step's camera forward reads: they run on a side stream and the
        # next encoder forward waits for them (same arithmetic, same order per parameter; PPV_OPT_OVERLAP=0 serialises).
        if opt_stream is not None:
            opt_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(opt_stream) if opt_stream is not None else contextlib.nullcontext():
            grads = [p.grad for p in enc_params]                                  # clip_gradient, train.py:311-316
            if not os.environ.get("PPV_BENCH_NOCLIP"):                            # (diagnostic only: is the optimiser stream on the critical path?)
                if hasattr(encoder, "clip_gradients_") and os.environ.get("PPV_BENCH_FLAT_CLIP", "1") != "0":
                    encoder.clip_gradients_(5.0)                                  # one pass over the flat gradient buffer (same values)
                else:
                    torch._foreach_clamp_min_(grads, -5.0)
                    torch._foreach_clamp_max_(grads, 5.0)
            opt_enc.step()
            if opt_stream is not None and hasattr(encoder, "prefetch_weight_layouts") and os.environ.get("PPV_WL_PREFETCH", "1") != "0":
                encoder.prefetch_weight_layouts()      # bf16 GEMM layouts of the updated weights, beside the next camera forward
            if decoder is not None:
                dgr = [p.grad for p in dec_params]
                torch._foreach_clamp_min_(dgr, -5.0)
                torch._foreach_clamp_max_(dgr, 5.0)
                opt_dec.step()
        return loss

    return step, enc_params + cam_params


def graph_step(step, device, warm=2):
    """Capture one whole training step (camera + encoder forward, losses, backward incl. its side streams, clipping, Adam, weight
    re-layout: ~800 launches) into a hipGraph and return a callable that replays it.  A dependent launch costs ~4.2 us on the
    command processor when enqueued one by one and ~1.6 us inside a graph (tools/micro/graph_chain.py); every replay executes the
    same kernels on the same (synthetic, resident) batch as the eager step."""
    s = torch.cuda.Stream(device=device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):                                   # allocator / lazy-init warm-up on the capture stream
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        loss = step()

    def replay():
        g.replay()
        return loss
    replay.graph = g
    return replay


def _latest_profile(suffix):
    """profiles/rNN<suffix> of the highest round that has one (the PMC passes cannot run inside this process: they are collected
    by tools/profile_round.sh on the same workload and committed)."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*" + suffix))):      # r02a < r02b < ...: the last snapshot of a round wins
        m = re.match(r"r(\d+)([a-z]*)", os.path.basename(f))
        if m and (best is None or (int(m.group(1)), m.group(2)) >= best[0]):
            best = ((int(m.group(1)), m.group(2)), f)
    return best[1] if best else None


def _load_profile(path):
    """-> (dict or None, stale).  A committed counter profile is used only if it was collected on the kernel sources the running
    library was built from (tools/csrc_hash.py, stored by tools/collect_pmc.py / collect_mfma.py as "csrc_sha16"); otherwise the
    counter-derived fields print as null with "stale": true."""
    try:
        d = json.load(open(path))
    except Exception:
        return None, False
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from csrc_hash import csrc_hash
    if d.get("csrc_sha16") != csrc_hash(ROOT):
        return None, True
    return d, False


_LIVE = {}          # "pmc": dict, "mfma": dict -- counter profiles collected in THIS run by live_counter_passes()


def live_counter_passes(budget_s=240):
    """The PMC counters of the roofline object measured in the driver's own run (they used to be read from committed
    JSON only).  Counters cannot be read inside this process, so rank 0 starts child runs of this script under rocprofv3 -- the program
    itself behind `--`, one counter group per pass as MI355X_MICROARCH.md prescribes, every launch serialised (PPV_WGRAD_SIDE=0): a
    kernel-trace pass for the durations, FETCH_SIZE, WRITE_SIZE and the MFMA-busy pass -- and folds their CSVs with tools/collect_pmc.py /
    collect_mfma.py.  Any failure (no rocprofv3, a pass over its time budget) leaves _LIVE empty: the line then falls back to the
    committed profiles and says so ("pmc_source")."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return
    t_start = time.perf_counter()
    tools = os.path.join(ROOT, "tools")
    td = tempfile.mkdtemp(prefix="ppv_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PPV_WGRAD_SIDE="0", PPV_BENCH_ONE_WINDOW="1")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-dense", "--no-roofline",
             "--no-configs", "--no-live-pmc"]
    passes = {"stats": ["--kernel-trace", "--stats"], "fetch": ["--pmc", "FETCH_SIZE"], "write": ["--pmc", "WRITE_SIZE"],
              "mfma": ["--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"]}
    found = {}
    try:
        for name, flags in passes.items():
            left = budget_s - (time.perf_counter() - t_start)
            if left < 20:
                return
            out = os.path.join(td, name)
            r = subprocess.run(["rocprofv3"] + flags + ["--output-format", "csv", "-d", out, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=left)
            pat = "*kernel_stats.csv" if name == "stats" else "*counter_collection.csv"
            hits = glob.glob(os.path.join(out, "**", pat), recursive=True)
            if r.returncode != 0 or not hits:
                print(f"[bench] live counter pass '{name}' failed (rc {r.returncode}): falling back to the committed profiles", file=sys.stderr, flush=True)
                return
            found[name] = hits[0]
        pj, mj = os.path.join(td, "pmc.json"), os.path.join(td, "mfma.json")
        a = subprocess.run([sys.executable, os.path.join(tools, "collect_pmc.py"), found["fetch"], found["write"], pj], capture_output=True, text=True)
        b = subprocess.run([sys.executable, os.path.join(tools, "collect_mfma.py"), found["mfma"], found["stats"], mj], capture_output=True, text=True)
        if a.returncode == 0 and b.returncode == 0:
            _LIVE["pmc"], _LIVE["mfma"] = json.load(open(pj)), json.load(open(mj))
            _LIVE["seconds"] = round(time.perf_counter() - t_start, 1)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] live counter passes: {e!r}: falling back to the committed profiles", file=sys.stderr, flush=True)
    finally:
        shutil.rmtree(td, ignore_errors=True)


def _step_hbm(sec_per_step):
    """Whole-step HBM view: bytes per step from the PMC passes (this run's live passes when they ran, else the committed ones of the
    same workload and kernel sources) over the measured step time."""
    if "pmc" in _LIVE:
        d = _LIVE["pmc"]
        gb = d["total_fetch_GB_per_step"] + d["total_write_GB_per_step"]
        tbs = gb / 1e3 / sec_per_step
        return {"GB_per_step_pmc": round(gb, 1), "pmc_source": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run", "TB_per_s": round(tbs, 2),
                "frac_of_8TBps_peak": round(tbs / 8.0, 3), "frac_of_6.3TBps_measured_copy_rate": round(tbs / 6.3, 3),
                "note": "the HEADLINE step (dense surface since round 6): ~6.8 GB of it are the 1.36-GB f32 output written, read by the head, and "
                        "its gradient written and pooled back; the lazy-consumer step moves that much less"}
    pmc = _latest_profile("_pmc_traffic.json")
    d, stale = _load_profile(pmc)
    if d is None:
        return {"GB_per_step_pmc": None, "stale": True, "pmc_file": os.path.basename(pmc)} if stale else None
    try:
        gb = d["total_fetch_GB_per_step"] + d["total_write_GB_per_step"]
    except Exception:
        return None
    tbs = gb / 1e3 / sec_per_step
    return {"GB_per_step_pmc": round(gb, 1), "pmc_file": os.path.basename(pmc), "TB_per_s": round(tbs, 2),
            "frac_of_8TBps_peak": round(tbs / 8.0, 3), "frac_of_6.3TBps_measured_copy_rate": round(tbs / 6.3, 3)}


# launch class (ppv_amd.convops labels) -> the kernel instantiations that serve it, as they are named in the rocprofv3 summaries
def _class_kernels(kind):
    tiled128 = lambda n: "conv_gemm_pipe_kernel<" in n and "conv_gemm_pipe_kernel<128, 64" not in n
    tiled64 = lambda n: "conv_gemm_pipe_kernel<128, 64" in n or "conv_gemm_kernel<64" in n
    return {
        "conv_gemm<128>": lambda n: tiled128(n) and n.endswith("false, false>"),
        "conv_gemm<128>+bn_sums": lambda n: tiled128(n) and n.endswith("false, true>"),
        "conv_gemm<64>": lambda n: tiled64(n) and n.endswith("false, false>"),
        "conv_gemm<64>+bn_sums": lambda n: tiled64(n) and n.endswith("false, true>"),
        "conv1x1_stream": lambda n: "conv1x1_stream_kernel" in n,
        "conv1x1_stream+bn_sums": lambda n: "conv1x1_stream_kernel" in n,
        "conv_wgrad": lambda n: "conv_wgrad" in n or "wgrad_to_torch" in n or "wgrad_reduce" in n,
    }.get(kind, lambda n: False)


def roofline_of_dominant_kernel(step):
    """One extra instrumented step: every conv launch is bracketed by HIP events on its own stream.  The DOMINANT class is the one
    with the most device time (not the most flops: that picks the fastest kernels)."""
    import ppv_amd.convops as co
    torch.cuda.synchronize()
    co.PROFILE = []
    # a launch's duration is its roofline input only if it has the device to itself: the instrumented step runs the weight
    # gradients on the main stream (in the timed steps they overlap the dgrad / BN chain on a side stream)
    prev = os.environ.get("PPV_WGRAD_SIDE")
    os.environ["PPV_WGRAD_SIDE"] = "0"
    try:
        step()
        torch.cuda.synchronize()
    finally:
        if prev is None:
            os.environ.pop("PPV_WGRAD_SIDE", None)
        else:
            os.environ["PPV_WGRAD_SIDE"] = prev
    rec, co.PROFILE = co.PROFILE, None
    agg = {}
    for kind, flops, e0, e1, nbytes in rec:
        a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += nbytes
    dom = max(agg, key=lambda k: agg[k][1])
    fl, sec, n, by = agg[dom]
    achieved = fl / sec / 1e12
    detail = {k: {"launches": v[2], "tflops": round(v[0] / v[1] / 1e12, 1), "frac_of_peak": round(v[0] / v[1] / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                  "ms": round(v[1] * 1e3, 3), "compulsory_TBps": round(v[3] / v[1] / 1e12, 2)} for k, v in agg.items()}
    all_fl, all_sec = sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())
    # the class's own roofline: arithmetic intensity against compulsory bytes (every operand and result once)
    ai = fl / by if by else None
    attainable = min(PEAK_BF16_DENSE_TFLOPS, ai * 8.0) if ai else None          # 8 TB/s HBM3E
    # HBM bytes per launch and MFMA-busy fraction of the dominant class from the latest committed PMC passes (rocprofv3 --pmc,
    # separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md); PMC cannot be collected inside this process
    match = _class_kernels(dom)
    traffic = mfma_busy = None
    pmc_file, mu_file = _latest_profile("_pmc_traffic.json"), _latest_profile("_mfma_util.json")
    if "pmc" in _LIVE:                                          # this run's own counter passes
        pmc_d, mu_d, stale = _LIVE["pmc"], _LIVE["mfma"], False
        pmc_file = mu_file = None
    else:
        pmc_d, stale_a = _load_profile(pmc_file)
        mu_d, stale_b = _load_profile(mu_file)
        stale = stale_a or stale_b
    try:
        pk = pmc_d["per_kernel"]
        tot_b = tot_n = 0.0
        for name, v in pk.items():
            if match(name):
                tot_b += (v["fetch_bytes_per_launch_corrected"] + v["write_bytes_per_launch"]) * v["launches_2steps"]
                tot_n += v["launches_2steps"]
        if dom == "conv_wgrad" and tot_n:                       # its slab-reduce launches belong to the same conv_wgrad() call
            tot_n = sum(v["launches_2steps"] for name, v in pk.items() if match(name) and "wgrad_to_torch" not in name and "wgrad_reduce" not in name)
        traffic = round(tot_b / tot_n) if tot_n else None
    except Exception:
        traffic = None
    try:
        pk = mu_d["per_kernel"]
        num = den = 0.0
        for name, v in pk.items():
            if match(name):
                num += v["mfma_busy_frac"] * v["avg_duration_us"] * v["launches"]
                den += v["avg_duration_us"] * v["launches"]
        mfma_busy = round(num / den, 4) if den else None
    except Exception:
        mfma_busy = None
    return {"bound": "mfma", "kernel": dom, "chosen_by": "largest share of device time among the MFMA launch classes",
            "mfma_busy_frac_pmc": mfma_busy, "pmc_files": [os.path.basename(f) for f in (pmc_file, mu_file) if f],
            "pmc_source": (f"live: four rocprofv3 child passes of this run ({_LIVE.get('seconds')} s)" if "pmc" in _LIVE else "committed profiles/ (hash-guarded)"),
            "stale": stale,   # true: the committed counter profiles were taken on other kernel sources -> traffic / mfma_busy are null
            "measured_in": "one extra step with every launch serialised on one stream (kernel alone on the device)", "achieved": round(achieved, 1), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic, "launches_per_step": n,
            "avg_launch_us": round(sec / n * 1e6, 2),
            "all_mfma_kernels": {"tflops": round(all_fl / all_sec / 1e12, 1), "frac_of_peak": round(all_fl / all_sec / 1e12 / PEAK_BF16_DENSE_TFLOPS, 4),
                                 "ms_per_step": round(all_sec * 1e3, 3), "algorithmic_TFLOP_per_step": round(all_fl / 1e12, 3)},
            "hbm_view": None if not by else {"compulsory_bytes_per_launch": round(by / n), "achieved_TBps": round(by / sec / 1e12, 2),
                                             "frac_of_8TBps": round(by / sec / 8e12, 3), "flop_per_byte": round(ai, 1),
                                             "attainable_TFLOPs_at_this_intensity": round(attainable, 1),
                                             "frac_of_attainable": round(achieved / attainable, 3)},
            "per_kernel": detail}


def cpu_baseline(camera):
    """The CPU oracle (oracle/: torch-CPU restatement of the reference camera + torch.nn ResNet-101) on B = 4 images."""
    from oracle import ic_camera as ic
    from oracle.resnet import Encoder as OEncoder
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                  # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    B = 4                                            # bounded sample: ~10-15 s of CPU work on 16 threads
    print(f"[bench] cpu_baseline: oracle on {cores} threads, B={B} ...", file=sys.stderr, flush=True)
    vol = camera.zernike_volume.cpu()
    coeffs = camera._concat().detach().cpu().requires_grad_(True)
    m1, m2 = ic.disk_masks()
    torch.manual_seed(2)
    enc = OEncoder()
    enc.train()
    img = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    noise = torch.rand(1, 896, 896, 1, generator=torch.Generator().manual_seed(1))

    def once():
        for p in enc.parameters():
            p.grad = None
        coeffs.grad = None
        sensor, psf, loss_psf = ic.forward(img, coeffs, vol, noise, prueba="3", mask_1=m1, mask_2=m2, height_tolerance=2e-8,
                                           sensor_distance=0.025, sample_interval=3e-6)
        out = enc(sensor)
        loss = 0.4 * (out * out).mean() + 6 * (1 - torch.nn.functional.mse_loss(img, sensor)) + 30 * loss_psf
        loss.backward()

    t0 = time.perf_counter()
    once()
    warm = time.perf_counter() - t0
    print(f"[bench] cpu_baseline warm-up {warm:.1f} s", file=sys.stderr, flush=True)
    ts = [warm]
    if warm < 10:                                   # one warm-up + three timed passes, median (bounded: <= ~30 s of CPU work)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            once()
            ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": round(B / t, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"camera (896/350/256, prueba '3') + ResNet-101 fwd+bwd fp32, B={B} @256x256, "
                      + (f"1 warm-up + {len(ts)} timed passes, median" if len(ts) > 1 else "single pass (warm-up took > 10 s)"),
            "passes_s": [round(x, 3) for x in ts]}


def side_config_legs(camera, encoder, batch, device):
    """BASELINE.json configs 2, 3 and 4 measured in the SAME run as the headline (VERDICT r4 task 3: the driver's one command then carries
    them): config 3 in-process on the headline's camera + encoder (10 timed steps with the caption decoder), configs 2 and 4 as child
    processes of their own scripts (tools/bench_camera.py with its own cpu_baseline, tools/bench_fd.py), one at a time."""
    import subprocess
    out = {}
    try:
        from ppv_amd.decoder import DecoderWithAttention
        torch.manual_seed(3)
        decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(device)
        decoder.train()
        step, _ = make_step(camera, encoder, batch, device, None, decoder, False, graph=False)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        wins = []
        for _ in range(3):                                        # three windows of 10 steps, the median one is reported (the first window
            t0 = time.perf_counter()                              # after a fresh decoder is 1-2 % slow: allocator growth, first launches)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t0) / 10)
        dt = sorted(wins)[1]
        out["3"] = {"metric": "images/sec fwd+bwd, Camera+ResNet-101+attention decoder @256^2, B=128, bf16 trunk / f32 decoder", "value": round(batch / dt, 1),
                    "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "steps": 10, "warmup": 3,
                    "windows_ms_per_step": [round(w * 1e3, 3) for w in wins],
                    "decoder_wgrad_path": os.environ.get("PPV_DEC_WGRAD", "default")}
        del step, decoder
    except Exception as e:  # noqa: BLE001
        out["3"] = {"error": repr(e)[:300]}
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    for key, script in (("2", "bench_camera.py"), ("4", "bench_fd.py")):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)], capture_output=True, text=True, timeout=240)
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[key] = json.loads(rows[-1]) if rows else {"error": (r.stderr or "no output")[-300:]}
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)[:300]}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `python -m torch.distributed.run` (one process per
    GPU, rendezvous on 127.0.0.1), relay its output (rank 0 prints the JSON line) and return its exit code.  Runs before anything in
    this process touches the GPU (torch.cuda.device_count() does not initialise it); never os.exec*."""
    import socket
    import subprocess
    visible = torch.cuda.device_count()
    if visible < n and not os.environ.get("PPV_FORCE_DEVICE0"):      # PPV_FORCE_DEVICE0=1: several ranks on one GPU (rehearsal)
        print(f"[bench] --gpus {n} but {visible} GPU(s) visible: refusing", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {n} ranks: {' '.join(cmd[1:10])} ...", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                                     # relay as it comes (rank 0's JSON line is the last one)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the secondary legs (the other output surface, the empty-queue host-enqueue "
                    "measurement): profiling runs execute headline steps only, so per-step counter totals divide cleanly")
    ap.add_argument("--lazy", action="store_true", help="diagnosis: make the lazy-output step (ppv_amd's own consumer) the headline instead "
                    "of the dense module surface")
    ap.add_argument("--decoder", action="store_true",
                    help="BASELINE.json config 3/5: add the attention decoder (512/512/512, 9490 words) to the step; the "
                         "default is the headline Camera+ResNet-101 metric")
    ap.add_argument("--ssim", action="store_true", help="camera_loss = 'SSIM' (fused SSIM kernels) instead of the default MSE")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not start the rocprofv3 child passes that measure roofline.traffic / "
                    "mfma_busy / step_hbm in this run (the committed profiles are used instead)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side legs of the default run (BASELINE.json configs 2, 3, 4 under "
                    "the line's \"configs\" key)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4],
                    help="BASELINE.json config: 0 = headline metric (default); 2 = camera alone (tools/bench_camera.py); 3 = same as "
                         "--decoder; 4 = FD camera + FAN + RAFT correlation (tools/bench_fd.py).  2 and 4 print their own JSON line.")
    args = ap.parse_args()
    if args.config in (2, 4):                                   # single-GPU side configurations: their own measured lines
        import runpy
        runpy.run_path(os.path.join(ROOT, "tools", "bench_camera.py" if args.config == 2 else "bench_fd.py"), run_name="__main__")
        return
    if args.config == 3:
        args.decoder = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                       # `python bench.py --gpus N`: N ranks as a child torchrun, nothing on the GPU here
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a mislabelled line",
              file=sys.stderr, flush=True)
        sys.exit(2)
    # PPV_FORCE_DIST=1: run the data-parallel machinery (process group, RCCL all-reduce on the side stream, global-max exchange)
    # even with one rank -- the one-GPU rehearsal of the N-GPU path (tests/test_dist_gpu.py)
    force_dist = world == 1 and bool(os.environ.get("PPV_FORCE_DIST"))
    if os.environ.get("PPV_FORCE_DEVICE0"):                          # rehearsal: several ranks on one GPU
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PPV_DIST_BACKEND", "nccl")        # "gloo" only for single-GPU rehearsals
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    rank = dist.get_rank() if world > 1 else 0

    torch.manual_seed(1234)                      # identical height-map noise stream on every rank -> identical PSF
    camera, encoder = build(device, global_max_sync=world > 1 or force_dist)
    sync = None
    if world > 1 or force_dist:
        from ppv_amd.dist_sync import GradSync
        sync = GradSync(bucket_mb=32)
        sync.defer_join = True                   # the step below flush()es before the optimisers: the tail bucket overlaps the camera's backward
        encoder.grad_sync = sync
    decoder = None
    if args.decoder:
        from ppv_amd.decoder import DecoderWithAttention
        torch.manual_seed(3)
        decoder = DecoderWithAttention(attention_dim=512, embed_dim=512, decoder_dim=512, vocab_size=9490, dropout=0.3).to(device)
        decoder.train()
    use_graph = os.environ.get("PPV_BENCH_GRAPH", "0") == "1"
    step, params = make_step(camera, encoder, args.batch, device, sync, decoder, args.ssim, graph=use_graph)
    eager_step = step
    if use_graph:
        step = graph_step(step, device)

    # The interpreter's cyclic garbage collector is parked for the warm-up + timed steps (everything the steps allocate is freed by
    # reference counting; a generation-2 sweep over the ~10^5 live objects of the modules takes 10-30 ms, i.e. 1-2 of the K = 20 timed
    # steps).  A training script gets the same with gc.freeze() after its set-up; the line records it ("gc").
    #
    # HEADLINE = the dense module surface (models.py:39-41: Encoder.forward writes its [B,E,E,2048] f32 tensor, the head reads it and hands
    # back a dense gradient); `value_lazy_consumer` = the same step with ppv_amd's own consumer (lazy dense tensor, head on the 8x8 cells).
    # Each is the MEDIAN of three windows of exactly K steps (`windows_ms_per_step`); --lazy swaps the two roles (diagnosis).
    import gc
    headline_dense = (not args.lazy) and decoder is None and hasattr(encoder, "lazy_output")
    n_win = 1 if (use_graph or args.no_dense or os.environ.get("PPV_BENCH_ONE_WINDOW")) else 3     # profiling runs: exactly W + K steps
    gc.collect()
    gc.freeze()
    gc.disable()
    try:
        with surface_mode(encoder, headline_dense):
            for _ in range(args.warmup):
                step()
            wins, enqs = timed_windows(step, args.steps, n_win, world, device)
        mid = sorted(range(n_win), key=lambda i: wins[i])[n_win // 2]
        elapsed, enqueued = wins[mid], enqs[mid]

        # Host cost of a step WITHOUT back-pressure: over the K timed steps the host runs ahead of the device until the command queue is
        # full and then enqueues at the device's pace, so `enqueued / K` converges to ms_per_step whenever the host is the faster
        # side.  Two steps into an empty queue measure the interpreter + runtime alone.
        torch.cuda.synchronize()
        host_free = None
        if not args.no_dense:
            with surface_mode(encoder, headline_dense):
                t1 = time.perf_counter()
                for _ in range(2):
                    step()
                host_free = (time.perf_counter() - t1) / 2
                torch.cuda.synchronize()

        # the other surface, same run, same protocol
        other = other_wins = None
        if decoder is None and not use_graph and not args.no_dense and hasattr(encoder, "lazy_output"):
            with surface_mode(encoder, not headline_dense):
                for _ in range(2):
                    step()
                other_wins, _ = timed_windows(step, args.steps, n_win, world, device)
            other = sorted(other_wins)[n_win // 2]
    finally:
        gc.enable()
        gc.unfreeze()

    if (rank == 0 and world == 1 and not args.no_live_pmc and not args.no_roofline and not args.no_dense and not args.decoder and not args.ssim
            and not use_graph and args.batch == 128 and os.environ.get("PPV_BENCH_LIVE_PMC", "1") != "0"):
        torch.cuda.synchronize()
        live_counter_passes()
    with surface_mode(encoder, headline_dense):
        roof = None if args.no_roofline else roofline_of_dominant_kernel(eager_step)
    side_configs = None
    if world == 1 and not args.decoder and not args.ssim and not use_graph and not args.no_configs and not args.no_dense and args.batch == 128:
        side_configs = side_config_legs(camera, encoder, args.batch, device)
    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        line = {
            "metric": "images/sec fwd+bwd, Camera+ResNet-101" + ("+attention decoder" if args.decoder else "") + " @256^2",
            "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "dist": ("rccl all-reduce exercised at world size 1" if force_dist else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "host_enqueue_ms_per_step": None if host_free is None else round(host_free * 1e3, 3),
            "host_enqueue_note": "interpreter + HIP runtime time to enqueue one step into an EMPTY queue (2 steps after a synchronise); "
                                 "host_enqueue_ms_per_step_timed_region is the same over the K timed steps, where a host that runs ahead "
                                 "is throttled by the full command queue",
            "host_enqueue_ms_per_step_timed_region": round(enqueued / args.steps * 1e3, 3),
            **surface_fields(world, args.batch, args.steps, wins, other_wins, headline_dense, decoder is not None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "IC OpticsZernike camera (896^2 wave grid, 350 Zernike terms, prueba '3') + ResNet-101 "
                                   "Encoder, fwd+bwd+Adam, 256x256; camera fp32/fp64, trunk bf16 storage + fp32 accumulate; "
                                   + ("" if args.decoder else ("the dense [B,36,36,2048] f32 output of models.py:39-41 is written, read by the stand-in head and "
                                      "its dense gradient pooled back INSIDE the timed step (value = value_dense_surface); value_lazy_consumer is the same "
                                      "step with ppv_amd's own consumer (lazy dense tensor, head on the 8x8 cells behind it), same protocol, same run; "
                                      if headline_dense else "LAZY dense output (--lazy diagnosis run): the head works on the 8x8 cells; "))
                                   + ("soft-attention LSTM decoder (512-d, 9490 words, captions of 9-18 tokens, CE + attention regulariser)"
                                      if args.decoder else "caption decoder not included (bench.py --decoder adds it)"),
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch,
                       "parallelism": f"dp{world}" if world > 1 else "single"},
            "trunk_mfma_frac_of_peak": round(value / world * TRUNK_GFLOP_PER_IMG * 1e9 / (PEAK_BF16_DENSE_TFLOPS * 1e12), 4),
            "roofline": roof,
            "step_hbm": _step_hbm(elapsed / args.steps) if (world == 1 and args.batch == 128 and not args.decoder and not args.ssim) else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(camera)
        if side_configs is not None:
            line["configs"] = side_configs
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
