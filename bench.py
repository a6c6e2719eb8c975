#!/usr/bin/env python3
"""bench.py — MIM train-step throughput (volumes/s) of the MI355X-native path, ViT-base, 512x512x320, B=1 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + loss + backward (+ RCCL gradient all-reduce when N > 1) + global-norm clip + AdamW + zero_grad on one
synthetic volume per GPU already resident in HBM (fp32 [1,320,1,512,512] U[0,1), mask from MaskGenerator(512,320,32,16,0.65):
BASELINE.json configs[1], the shape README.md:56-68 of the reference recommends).  Weights: random init of the ViT-base
VideoMAE-for-pretraining architecture.  Rank 0 prints ONE JSON line (see the task contract): value = volumes/s over all GPUs.

Extra objects on that line:
  roofline      dominant kernel (decoder attention backward dK/dV, S=20,480, 6 heads): algorithmic FLOP per launch / average
                launch duration measured with HIP events on the launch stream inside this process, vs the 2.5 PFLOP/s dense
                bf16 MFMA peak;  `step` adds the whole-step figure (18.46 TFLOP algorithmic fwd+bwd per volume).
  cpu_baseline  the CPU oracle (oracle/mim_oracle.py, kind "port") timed on this host's cores on a bounded sample
                (one encoder layer + one decoder layer at full sequence length + embedding/head/loss, fwd+bwd, extrapolated to
                12 + 4 layers).  Rank 0, N == 1 only.  Skipped with --no-cpu-baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
STEP_TFLOP = 18.461           # algorithmic fwd+bwd per volume

VIT_BASE = dict(image_size=512, num_frames=320, patch_size=16, tubelet_size=16, num_channels=1, hidden_size=768, num_hidden_layers=12,
                num_attention_heads=12, intermediate_size=3072, decoder_hidden_size=384, decoder_num_hidden_layers=4,
                decoder_num_attention_heads=6, decoder_intermediate_size=1536)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build(rank, device, cfg_kw):
    from transformers import VideoMAEConfig
    from smb_vision_amd.masking import MaskGenerator
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    torch.manual_seed(42)
    model = SmbVideoMAEForPreTraining(VideoMAEConfig(**cfg_kw)).to(device).train()
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    pix = torch.rand(1, cfg_kw["num_frames"], 1, cfg_kw["image_size"], cfg_kw["image_size"], generator=g, device=device)
    np.random.seed(42 + rank)
    gen = MaskGenerator(input_size=cfg_kw["image_size"], depth=cfg_kw["num_frames"], mask_patch_size=32, model_patch_size=cfg_kw["patch_size"],
                        mask_ratio=0.65)
    mask = gen().unsqueeze(0).to(device)
    return model, pix, mask, gen.num_masked


def kernel_roofline(device, iters=20, warm=10):
    """Decoder attention backward dK/dV kernel at the bench shape, timed with HIP events on the stream it is launched on."""
    from smb_vision_amd import ops
    B, S, H = 1, 20480, 6
    w = H * 64
    g = torch.Generator(device=device).manual_seed(0)
    qkv = torch.randn(B * S, 3 * w, generator=g, device=device).bfloat16()
    dO = torch.randn(B * S, w, generator=g, device=device).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H)
    out = (torch.empty_like(qkv), torch.empty(B, H, S, dtype=torch.float32, device=device))
    ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=1, out=out)
    res = {}
    for name, parts, flop in (("attn_dkv_kernel", 2, 4 * 2.0 * S * S * 64 * H), ("attn_dq_kernel", 4, 3 * 2.0 * S * S * 64 * H)):
        for _ in range(warm):  # the host-side set-up above leaves the GPU idle long enough for its clocks to drop
            ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=parts, out=out)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(iters):
            ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=parts, out=out)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / iters
        res[name] = dict(ms=ms, tflops=flop / (ms * 1e-3) / 1e12, flop=flop)
    return res


def cpu_threads():
    """Host threads for the CPU leg: the cgroup CPU quota when there is one (a 1-GPU box gets a 16-CPU share of a 256-thread
    host; running 256 threads inside it is 2-3x slower), else the affinity mask, capped at 64."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    env = os.environ.get("SMBV_CPU_THREADS")
    return int(env) if env else min(n, 64) if n <= 64 else 16


def cpu_baseline(threads):
    """Oracle (port) timed on the host: bounded sample, extrapolated (see module docstring)."""
    import oracle.mim_oracle as o
    torch.set_num_threads(threads)
    o.SDPA = True
    cfg = o.MIMConfig()
    g = torch.Generator().manual_seed(0)

    def layer_time(S, w, ff, heads):
        c = o.MIMConfig(hidden_size=w, intermediate_size=ff, num_attention_heads=heads)
        names = o.layer_param_names("l.")
        shapes = [(w,), (w,), (w, w), (w, w), (w, w), (w,), (w,), (w, w), (w,), (w,), (w,), (ff, w), (ff,), (w, ff), (w,)]
        p = {n: (torch.randn(s, generator=g) * 0.02).requires_grad_(True) for n, s in zip(names, shapes)}
        x = torch.randn(1, S, w, generator=g).requires_grad_(True)
        t0 = time.perf_counter()
        y = o.layer(x, p, "l.", heads, c.layer_norm_eps)
        y.sum().backward()
        return time.perf_counter() - t0

    t_enc = layer_time(7168, 768, 3072, 12)
    t_dec = layer_time(20480, 384, 1536, 6)
    # embedding (all N tokens, as the reference computes it) + head + targets + loss
    pix = torch.rand(1, 320, 1, 512, 512, generator=g)
    w_pe = (torch.randn(768, 1, 16, 16, 16, generator=g) * 0.02).requires_grad_(True)
    b_pe = torch.zeros(768, requires_grad=True)
    w_h = (torch.randn(4096, 384, generator=g) * 0.02).requires_grad_(True)
    h = torch.randn(1, 13312, 384, generator=g).requires_grad_(True)
    t0 = time.perf_counter()
    emb = o.patch_embed(pix, w_pe, b_pe, cfg)
    logits = torch.nn.functional.linear(h, w_h)
    labels = o.labels_from_pixels(pix, cfg)[:, :13312]
    loss = torch.nn.functional.mse_loss(logits, labels) + emb.mean()
    loss.backward()
    t_rest = time.perf_counter() - t0
    est = 12 * t_enc + 4 * t_dec + t_rest
    return dict(value=1.0 / est, unit="volumes/s", cores=threads, kind="port",
                sample=f"oracle fp32 fwd+bwd: 1 of 12 encoder layers at S=7168 ({t_enc:.2f}s) + 1 of 4 decoder layers at S=20480 "
                       f"({t_dec:.2f}s) + patch-embed/head/targets/loss ({t_rest:.2f}s), extrapolated to the full step ({est:.1f}s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward only (diagnostic; not the metric)")
    ap.add_argument("--torch-optimizer", action="store_true", help="clip_grad_norm_ + torch fused AdamW instead of the fused HIP optimizer tail")
    ap.add_argument("--tiny", action="store_true", help="64x64x32 smoke shape (diagnostic; not the metric)")
    ap.add_argument("--vit-large", action="store_true", help="BASELINE.json config 5's model in bf16 (encoder 1024/24L/16H/4096; decoder 512/12L/8H/2048 - the reference "
                                                               "specifies none, SURVEY.md 8d) - diagnostic; not the metric")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse the N>1 path on one GPU)")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MIM hot path has no CPU fallback")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    cfg_kw = dict(VIT_BASE)
    if args.tiny:
        cfg_kw.update(image_size=64, num_frames=32, hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=512,
                      decoder_hidden_size=64, decoder_num_attention_heads=1, decoder_num_hidden_layers=1, decoder_intermediate_size=256)
    if args.vit_large:
        cfg_kw.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, decoder_hidden_size=512,
                      decoder_num_hidden_layers=12, decoder_num_attention_heads=8, decoder_intermediate_size=2048)
    model, pix, mask, n_masked = build(rank, device, cfg_kw)
    params = [p for p in model.parameters()]
    opt = None
    if not args.no_optimizer:
        if args.torch_optimizer:
            opt = torch.optim.AdamW(params, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, fused=True)
        else:  # fused global-norm clip + AdamW in libsmbv.so (same arithmetic as clip_grad_norm_ + torch AdamW; tests/test_ops_gpu.py)
            from smb_vision_amd.optim import SmbvAdamW
            opt = SmbvAdamW(params, lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
    dp = None
    if world > 1:
        from smb_vision_amd.dp import DataParallelEngine
        dp = DataParallelEngine(model)
        dp.broadcast_parameters(model)

    def step():
        out = model(pixel_values=pix, bool_masked_pos=mask, num_masked=n_masked)
        out.loss.backward()
        if opt is not None:
            if args.torch_optimizer:
                torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
            opt.step()
            opt.zero_grad(set_to_none=True)
        else:
            model.zero_grad(set_to_none=True)
        return out.loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    final_loss = float(loss.detach())
    in_sync = None
    if world > 1:  # replicas must hold identical parameters after K averaged steps (outside the timed region)
        cs = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool((hi - lo).abs().item() <= 1e-9 * max(1.0, abs(float(hi))))

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.steps / dt
        line = {
            "metric": "MIM train-step volumes/sec (512x512x320, ViT-base)", "value": value, "unit": "volumes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ViT-base MIM, 512x512x320 volume, patch_size=16, mask_ratio=0.65 (mask_patch 32), B=1/GPU, "
                                   "fwd+loss+bwd+clip+AdamW" + ("" if opt is not None else " [NO OPTIMIZER: diagnostic]")
                                   + (" [TINY: diagnostic]" if args.tiny else "") + (" [ViT-LARGE bf16: diagnostic, 64.41 TFLOP/volume]" if args.vit_large else ""),
                       "tokens": model.num_patches, "masked": n_masked, "parallelism": f"dp{world}", "final_loss": final_loss,
                       "replicas_in_sync": in_sync},
        }
        step_flop = 64.41 if args.vit_large else STEP_TFLOP
        step_tf = world * step_flop / (ms * 1e-3) if not args.tiny else None  # whole job, TFLOP/s
        if world > 1 and not args.tiny:  # multi-GPU lines carry the whole-step roofline only (the kernel figure is a 1-GPU property)
            line["roofline"] = {"bound": "mfma", "kernel": "whole step (per GPU)", "achieved": step_tf / world, "peak": MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": step_tf / world / MFMA_PEAK_TFLOPS, "traffic": None}
        elif not args.no_kernel_roofline and not args.tiny:
            kr = kernel_roofline(device)
            dom = kr["attn_dkv_kernel"]
            traffic = None
            try:  # PMC-measured fabric bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes), committed under profiles/
                traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_attn_dkv_traffic.json")))["traffic_bytes_per_launch"]
            except Exception:
                pass
            line["roofline"] = {"bound": "mfma", "kernel": "attn_dkv2_kernel (decoder, S=20480, H=6, Dh=64)", "achieved": dom["tflops"],
                                "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic,
                                "launch_ms": dom["ms"], "flop_per_launch": dom["flop"],
                                "other": {"attn_dq_kernel": {"launch_ms": kr["attn_dq_kernel"]["ms"], "achieved": kr["attn_dq_kernel"]["tflops"]}},
                                "step": {"achieved": step_tf, "flop_per_volume_T": step_flop, "frac": step_tf / MFMA_PEAK_TFLOPS}}
        if world == 1 and not args.no_cpu_baseline and not args.tiny:
            try:
                line["cpu_baseline"] = cpu_baseline(cpu_threads())
            except Exception as e:  # the baseline leg must never take the bench line down
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
