#!/usr/bin/env python3
"""bench.py — MIM train-step throughput (volumes/s) of the MI355X-native path, ViT-base, 512x512x320, B=1 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1 with no launcher: bench.py starts its N ranks itself, as child processes - self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + loss + backward (+ RCCL gradient all-reduce when N > 1) + global-norm clip + AdamW + zero_grad on one synthetic
volume per GPU (fp32 [1,320,1,512,512] U[0,1), mask from MaskGenerator(512,320,32,16,0.65): BASELINE.json configs[1], the shape
README.md:56-68 of the reference recommends).  Weights: random init of the ViT-base VideoMAE-for-pretraining architecture.  Rank 0
prints ONE JSON line (the task contract):

  value / ms_per_step   the contract's number: K steps on a volume ALREADY RESIDENT in HBM, between barriers, max over ranks.
  fed                   the same K steps with SURVEY.md §8(d)'s full step: every step takes a NEW raw int16 volume from page-locked host
                        memory through smb_vision_amd.feed.VolumeFeeder (copy stream + device-side ScaleIntensityRange / pad / crop / permute) and
                        a NEW mask from the bit-exact MaskGenerator with host-built token lists; same barrier protocol.  PCIe-inclusive, therefore
                        reported beside `value`, never as it.
  roofline              dominant kernel = the decoder attention backward (attn_bwd_pair_kernel: key-owner and query-owner blocks in ONE launch, S = 20,480, 6 heads),
                        timed with HIP events on its launch stream: `frac` on SURVEY.md 8(d)'s count (backward = 2 x forward = FOUR S^2 Dh H products, the convention the
                        step figure uses too), `frac_5_products` on the five a fused flash backward executes (S, dP, dV, dK, dQ), `executed_frac` on the seven
                        this launch executes (its query-owner blocks recompute S and dP); `dkv` is the dK/dV kernel alone;
                        `step` the whole-step figure (18.46 TFLOP algorithmic fwd+bwd per volume).  Peak 2.5 PFLOP/s dense bf16.
  cpu_baseline          the CPU oracle (oracle/mim_oracle.py, kind "port", fused CPU attention as the reference defaults to SDPA) on this host's cores:
                        one warm-up step on a 2+1-layer model, then 2 timed FULL steps (forward + loss + backward) of the same workload.  Rank 0, N == 1.
  cpu_baseline.reference_class   the class the reference trains (src/run_mim.py:13-21 imports transformers.VideoMAEForPreTraining) on the same workload and cores:
                        fp32, SDPA attention, forward + loss + backward, one warm-up on a 2+1-layer model and ONE timed full step - beside the port, which stays the
                        headline CPU figure (kind "port"); it validates that the port is not a slower or faster program than the reference's own class.
  config.vjepa_ms_per_step / config.vit_large_fp8_recompute_ms_per_step   BASELINE.json configs 4 and 5 (one GPU), a few steps each after the headline leg.
  config.no_optimizer_ms_per_step   forward + loss + backward only ("with and without optimizer"); config.b4_ms_per_volume: four volumes per step
                        (the reference's shipped recipe, scripts/training/run_mim.sh:13-38) on the same model, per volume.
  config.trainer_ms_per_step   the same workload through the shipped harness (smb_vision_amd.run_mim: HF Trainer + SmbTrainer's optimizer tail) with
                        volumes handed over as pinned fp32 CPU tensors the way the reference's DataLoader does; N == 1 only.
N > 1: `fed` is reported too (every rank feeds its own volumes over PCIe at once).
Diagnostics (not the metric): --vjepa (BASELINE.json config 4), --vit-large (config 5's model), --tiny, --no-optimizer, --torch-optimizer.
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
VJEPA_STEP_TFLOP = 102.2      # 3 x encoder + 3 x predictor + 1 x target encoder

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
    return model, pix, mask, gen


def kernel_roofline(device, iters=20, warm=10):
    """Decoder attention backward at the bench shape, timed with HIP events on the stream it is launched on: the dK/dV kernel alone, the dQ
    kernel alone, and the pair as the step runs it (one launch with the two kinds of blocks mixed per CU, attention_bwd.hip)."""
    from smb_vision_amd import ops
    B, S, H = 1, 20480, 6
    w = H * 64
    g = torch.Generator(device=device).manual_seed(0)
    qkv = torch.randn(B * S, 3 * w, generator=g, device=device).bfloat16()
    dO = torch.randn(B * S, w, generator=g, device=device).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H)
    out = (torch.empty_like(qkv), torch.empty(B, H, S, dtype=torch.float32, device=device))
    ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=1, out=out)
    unit = 2.0 * S * S * 64 * H  # one S x S x 64 product over all heads
    res = {}
    for name, parts in (("dkv", 2), ("dq", 4), ("pair", 6)):
        for _ in range(warm):  # the host-side set-up above leaves the GPU idle long enough for its clocks to drop
            ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=parts, out=out)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(iters):
            ops.attention_bwd(qkv, O, dO, lse, B, S, H, parts=parts, out=out)
        ev1.record()
        torch.cuda.synchronize()
        res[name] = ev0.elapsed_time(ev1) / iters
    return res, unit


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(threads, timed=2):
    """The oracle's whole train step (forward + loss + backward, fp32, fused CPU attention) on the bench workload, on the host's cores."""
    import oracle.mim_oracle as o
    from smb_vision_amd.masking import MaskGenerator
    torch.set_num_threads(threads)
    o.SDPA = True
    try:
        kw = {k: v for k, v in VIT_BASE.items() if k != "num_channels"}
        np.random.seed(42)
        mask = MaskGenerator(512, 320, 32, 16, 0.65)().unsqueeze(0)
        pix = torch.rand(1, 320, 1, 512, 512, generator=torch.Generator().manual_seed(1234))
        small = o.MIMConfig(**dict(kw, num_hidden_layers=2, decoder_num_hidden_layers=1))
        o.mim_step(o.init_params(small, seed=42), small, pix, mask)  # warm-up: thread pool, allocator, kernels selected
        cfg = o.MIMConfig(**kw)
        params = o.init_params(cfg, seed=42)
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            loss, _, _ = o.mim_step(params, cfg, pix, mask)
            ts.append(time.perf_counter() - t0)
    finally:
        o.SDPA = False
    mean = sum(ts) / len(ts)
    return dict(value=1.0 / mean, unit="volumes/s", cores=threads, kind="port", cpu=cpu_model(), seconds_per_step=[round(t, 2) for t in ts],
                loss=float(loss),
                sample=f"oracle/mim_oracle.py mim_step (fp32, SDPA attention): forward + loss + backward of the whole ViT-base 512x512x320 workload, 1 warm-up "
                       f"step on a 2+1-layer model then {timed} timed full steps ({', '.join(f'{t:.1f}s' for t in ts)}), optimizer excluded")


def reference_class_baseline(threads):
    """transformers.VideoMAEForPreTraining - the class src/run_mim.py:13-21 imports and :345-357 instantiates - on the bench workload: fp32, SDPA,
    forward + loss + backward on the host's cores.  One warm-up on a 2+1-layer model, one timed full step."""
    from transformers import VideoMAEConfig, VideoMAEForPreTraining
    from smb_vision_amd.masking import MaskGenerator
    torch.set_num_threads(threads)
    np.random.seed(42)
    mask = MaskGenerator(512, 320, 32, 16, 0.65)().unsqueeze(0)
    pix = torch.rand(1, 320, 1, 512, 512, generator=torch.Generator().manual_seed(1234))

    def one(cfg_kw, bf16=False):
        cfg = VideoMAEConfig(**cfg_kw)
        cfg._attn_implementation = "sdpa"
        torch.manual_seed(42)
        m = VideoMAEForPreTraining(cfg).float().train()
        t0 = time.perf_counter()
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bf16):
            out = m(pixel_values=pix, bool_masked_pos=mask)
        out.loss.backward()
        return time.perf_counter() - t0, float(out.loss.detach())

    one(dict(VIT_BASE, num_hidden_layers=2, decoder_num_hidden_layers=1))
    dt, loss = one(VIT_BASE)
    res = dict(value=1.0 / dt, unit="volumes/s", seconds_per_step=round(dt, 2), loss=loss, cores=threads,
               what="transformers.VideoMAEForPreTraining (the reference's model class, src/run_mim.py:13-21), fp32, sdpa, forward + loss + backward, 1 timed step")
    try:  # "fp32 and bf16-autocast" - the precision policy the reference trains under (--bf16 true), on the same cores
        one(dict(VIT_BASE, num_hidden_layers=2, decoder_num_hidden_layers=1), bf16=True)
        dtb, lossb = one(VIT_BASE, bf16=True)
        res["bf16_autocast"] = dict(value=1.0 / dtb, unit="volumes/s", seconds_per_step=round(dtb, 2), loss=lossb,
                                    what="the same class and step under torch.autocast('cpu', bfloat16), 1 timed step")
    except Exception as e:
        res["bf16_autocast"] = {"error": repr(e)}
    return res


class _FedSource:
    """Endless (raw int16 HU volume in page-locked memory, extras) stream: 4 distinct volumes cycled, a fresh mask per step from the bit-exact
    generator (global numpy RNG, as the reference's GenerateMask) with the token lists built on the host."""

    def __init__(self, rank, gen, size, depth):
        from smb_vision_amd.masking import split_mask_indices_host
        self.split = split_mask_indices_host
        rng = np.random.default_rng(1234 + rank)
        self.raws = [torch.from_numpy(rng.integers(-1000, 1000, size=(size, size, depth), dtype=np.int16)).pin_memory() for _ in range(4)]
        self.gen = gen

    def __iter__(self):
        i = 0
        while True:
            mask = self.gen().unsqueeze(0)
            vis, msk = self.split(mask)
            yield self.raws[i & 3], dict(mask=mask.pin_memory(), vis=vis.pin_memory(), msk=msk.pin_memory())
            i += 1


def trainer_ms_per_step(cfg_kw, steps=10, warm=8):
    """The shipped harness at full size: smb_vision_amd.run_mim.build_trainer (HF Trainer + SmbTrainer), B = 1, data = pinned fp32 CPU volumes.
    (8 warm-up steps: the first upload out of each of the 4 freshly page-locked volumes is slow, as in the fed leg.)"""
    import tempfile
    from transformers import TrainerCallback
    from smb_vision_amd import run_mim

    class Cycling(torch.utils.data.Dataset):
        def __init__(self, gen):
            self.vols = [torch.rand(cfg_kw["num_frames"], 1, cfg_kw["image_size"], cfg_kw["image_size"], generator=torch.Generator().manual_seed(s)).pin_memory()
                         for s in range(4)]
            self.gen = gen

        def __len__(self):
            return 4 * (steps + warm)

        def __getitem__(self, i):
            return {"image": self.vols[i & 3], "mask": self.gen()}

    class Clock(TrainerCallback):
        t0 = t1 = None

        def on_step_end(self, args, state, control, **kw):
            if state.global_step == warm:
                torch.cuda.synchronize()
                Clock.t0 = time.perf_counter()
            if state.global_step == warm + steps:
                torch.cuda.synchronize()
                Clock.t1 = time.perf_counter()

    import contextlib
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(sys.stderr):  # stdout carries the ONE JSON line only
        argv = ["--output_dir", tmp, "--image_size", str(cfg_kw["image_size"]), "--depth", str(cfg_kw["num_frames"]), "--patch_size", "16",
                "--mask_patch_size", "32", "--mask_ratio", "0.65", "--max_steps", str(warm + steps), "--per_device_train_batch_size", "1",
                "--report_to", "none", "--bf16", "true", "--learning_rate", "1.5e-4", "--weight_decay", "0.05", "--adam_beta2", "0.95", "--logging_steps", "1000",
                "--save_strategy", "no", "--max_grad_norm", "1.0", "--dataloader_num_workers", "0", "--dataloader_pin_memory", "false", "--gradient_checkpointing", "true",
                "--decoder_hidden_size", "384", "--decoder_num_hidden_layers", "4", "--decoder_num_attention_heads", "6", "--decoder_intermediate_size", "1536",
                "--disable_tqdm", "true"]
        trainer = run_mim.build_trainer(argv)
        trainer.train_dataset = Cycling(trainer.train_dataset.gen)
        trainer.add_callback(Clock())
        trainer.train()
    del trainer
    torch.cuda.empty_cache()
    return (Clock.t1 - Clock.t0) / steps * 1e3


def vjepa_leg(device, world, rank, steps, warmup):
    """BASELINE.json config 4 (ViT-base encoder + default predictor, 512x512x320, B = 1, 3 mask blocks): online forward + frozen target
    forward + L1 + EMA + backward + fused clip / AdamW.  Returns (seconds for `steps` steps, final loss, info)."""
    from transformers import VJEPA2Config
    from smb_vision_amd.masking import VJEPAMaskGenerator
    from smb_vision_amd.modeling_vjepa import MomentumEncoder, SmbVJEPA2Model, make_target_encoder, vjepa_loss
    from smb_vision_amd.optim import SmbvAdamW
    cfg = VJEPA2Config(patch_size=16, crop_size=512, frames_per_clip=320, tubelet_size=16, in_chans=1, hidden_size=768, num_hidden_layers=12,
                       num_attention_heads=12, mlp_ratio=4.0, pred_hidden_size=384, pred_num_hidden_layers=12, pred_num_attention_heads=12)
    torch.manual_seed(0)
    model = SmbVJEPA2Model(cfg).to(device).train()
    target = make_target_encoder(model)
    ema = MomentumEncoder(target, 0.99925)
    opt = SmbvAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
    if world > 1:
        from smb_vision_amd.dp import DataParallelEngine
        DataParallelEngine(model).broadcast_parameters(model)
    pix = torch.rand(1, 320, 1, 512, 512, generator=torch.Generator(device=device).manual_seed(1 + rank), device=device)
    torch.manual_seed(7 + rank)
    item = VJEPAMaskGenerator(input_size=(320, 512, 512), patch_size=(16, 16, 16), pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=3)({})
    # what run_vjepa.collate_fn hands over next to the masks: the step's index arrays, built on the host (the DataLoader's workers in the harness)
    from smb_vision_amd.modeling_vjepa import index_plumbing
    cm_h, tm_h = item["context_mask"][None].long(), item["target_mask"][None].long()
    to_dev = lambda d: {k: (v.to(device) if torch.is_tensor(v) else ([t.to(device) for t in v] if isinstance(v, list) else v)) for k, v in d.items()}
    cm, tm = cm_h.to(device), tm_h.to(device)
    inputs = {"pixel_values_videos": pix, "context_mask": [cm], "target_mask": [tm]}
    if os.environ.get("SMBV_VJEPA_DEVICE_PLUMBING", "0") != "1":
        inputs["smbv_plumbing_context"] = to_dev(index_plumbing([cm_h], [tm_h], model.num_patches, "context"))
        inputs["smbv_plumbing_target"] = to_dev(index_plumbing([cm_h], [tm_h], model.num_patches, "target"))

    def step():
        loss, _ = vjepa_loss(model, target, ema, inputs)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(warmup):
        loss = step()
    dt = timed_region(step, steps, world, device)
    info = dict(tokens=model.num_patches, context=int(cm.shape[1]), target=int(tm.shape[1]), final_loss=float(loss.detach()))
    del model, target, ema, opt
    torch.cuda.empty_cache()
    return dt, info


def run_vjepa(args, device, world, rank):
    """--vjepa: the config-4 leg as the bench's only line (diagnostic), same barrier protocol."""
    dt, info = vjepa_leg(device, world, rank, args.steps, args.warmup)
    if rank == 0:
        ms = dt / args.steps * 1e3
        tf = world * VJEPA_STEP_TFLOP / (ms * 1e-3)
        print(json.dumps({"metric": "V-JEPA train-step volumes/sec (512x512x320, ViT-base + predictor) [diagnostic: BASELINE.json config 4]", "value": world * args.steps / dt,
                          "unit": "volumes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "V-JEPA ViT-base encoder (12L/768) + predictor (12L/384, 12 heads x 32), 512x512x320, B=1/GPU, 3 mask blocks: online fwd + "
                                                 "target fwd + L1 + EMA + bwd + clip + AdamW", "tokens": info["tokens"], "context": info["context"], "target": info["target"],
                                     "parallelism": f"dp{world}", "final_loss": info["final_loss"]},
                          "roofline": {"bound": "mfma", "kernel": "whole step (per GPU)", "achieved": tf / world, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf / world / MFMA_PEAK_TFLOPS, "traffic": None, "flop_per_volume_T": VJEPA_STEP_TFLOP}}), flush=True)


def vit_large_leg(device, steps=4, warmup=2):
    """BASELINE.json config 5's model on one GPU: ViT-large MIM (encoder 1024/24L/16H/4096, decoder 512/12L/8H/2048), fp8 attention forward + per-layer
    recompute; fwd + loss + bwd + clip + AdamW.  Returns ms per step."""
    from smb_vision_amd.optim import SmbvAdamW
    kw = dict(VIT_BASE, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, decoder_hidden_size=512,
              decoder_num_hidden_layers=12, decoder_num_attention_heads=8, decoder_intermediate_size=2048, smbv_attention_fp8="all", smbv_recompute="always")
    model, pix, mask, _ = build(0, device, kw)
    model.gradient_checkpointing_enable()
    opt = SmbvAdamW(list(model.parameters()), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)

    def step():
        model(pixel_values=pix, bool_masked_pos=mask).loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(warmup):
        step()
    dt = timed_region(step, steps, 1, device)
    del model, opt
    torch.cuda.empty_cache()
    return dt / steps * 1e3


def b4_leg(model, opt, gen, pix, device, steps=5, warmup=2):
    """Four volumes per step on the headline model (the shipped launch recipe's batch, scripts/training/run_mim.sh:13-38): fwd + loss + bwd + clip +
    AdamW; returns (ms per VOLUME, info).  Volumes 2-4 are rolled copies of the resident one (distinct data, no 1 GB of new random numbers), each with
    its own mask and host-built token lists."""
    from smb_vision_amd.masking import split_mask_indices_host
    B = 4
    pix4 = torch.cat([pix] + [torch.roll(pix, shifts=17 * i, dims=-1) for i in range(1, B)])
    mask4 = torch.stack([gen() for _ in range(B)])
    vis, msk = split_mask_indices_host(mask4)
    mask4, vis, msk = mask4.to(device), vis.to(device), msk.to(device)
    last = {}

    def step():
        out = model(pixel_values=pix4, bool_masked_pos=mask4, visible_index=vis, masked_index=msk)
        out.loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        last["loss"] = out.loss

    for _ in range(warmup):
        step()
    dt = timed_region(step, steps, 1, device)
    ms = dt / steps * 1e3
    info = dict(ms_per_step=ms, volumes_per_s=B * steps / dt, frac_of_mfma_peak=B * STEP_TFLOP / (ms * 1e-3) / MFMA_PEAK_TFLOPS, final_loss=float(last["loss"].detach()),
                what="B = 4 volumes per step on one GPU (the reference's shipped recipe, scripts/training/run_mim.sh:13-38), same model / optimizer as the headline leg")
    del pix4
    model._ws = None   # the B = 4 workspace (~21 GB) is not kept for the legs that follow
    torch.cuda.empty_cache()
    return ms / B, info


def reference_class_on_gpu(device, steps=4, warmup=4):
    """The reference's model class on the SAME GPU, for scale (a diagnostic beside the metric; nothing of it is on the product path):
    transformers.VideoMAEForPreTraining (src/run_mim.py:13-21) on the bench workload under bf16 autocast (the Trainer's --bf16) with torch clip + fused
    AdamW, once with PyTorch-ROCm's own scaled_dot_product_attention ("sdpa": what the reference stack runs on an MI355X; flash-attn is not installed)
    and once with only the attention swapped for the HIP kernels through transformers' attention registry ("smbv", smb_vision_amd/attention_plugin.py).
    (Four warm-up steps: MIOpen benchmarks its Conv3d candidates - 113-ms naive kernels among them - during the class's first calls; with two warm-up steps that search
    reached into the timed region and the leg read 56-71 ms for 'smbv' instead of 47, tools/r05_plugin_after_trainer.py.)"""
    from transformers import VideoMAEConfig, VideoMAEForPreTraining
    from smb_vision_amd.masking import MaskGenerator
    from smb_vision_amd import attention_plugin
    attention_plugin.register()
    np.random.seed(42)
    mask = MaskGenerator(512, 320, 32, 16, 0.65)().unsqueeze(0).to(device)
    pix = torch.rand(1, 320, 1, 512, 512, generator=torch.Generator().manual_seed(1234)).to(device)
    res = {}
    for impl in ("sdpa", attention_plugin.NAME):
        torch.manual_seed(42)
        m = VideoMAEForPreTraining(VideoMAEConfig(**VIT_BASE, attn_implementation=impl)).to(device).train()
        opt = torch.optim.AdamW(m.parameters(), lr=1.5e-4, betas=(0.9, 0.95), weight_decay=0.05, fused=True)
        last = {}

        def step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = m(pixel_values=pix, bool_masked_pos=mask)
            out.loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()
            opt.zero_grad(set_to_none=True)
            last["loss"] = out.loss.detach()

        for _ in range(warmup):
            step()
        dt = timed_region(step, steps, 1, device)
        res[impl] = dict(ms_per_step=dt / steps * 1e3, loss=float(last["loss"]))
        del m, opt
        torch.cuda.empty_cache()
    res["what"] = ("transformers.VideoMAEForPreTraining on this GPU, bf16 autocast, fwd + loss + bwd + torch clip + fused AdamW: 'sdpa' = PyTorch-ROCm's "
                   "attention, 'smbv' = the same class with only the attention swapped for the HIP kernels via the attention registry")
    return res


def timed_region(step, steps, world, device):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; returns the max over ranks of the elapsed seconds."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` as the driver runs it (no launcher, WORLD_SIZE unset): start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process
    (one rank per GPU, replacing `accelerate launch --config_file acc_configs/multi_gpu_config.yaml`, scripts/training/run_vjepa.sh:16), relay rank 0's JSON line
    on stdout (everything else the ranks print goes to stderr) and return the launcher's exit code (non-zero when any rank failed).  The parent never touches
    the GPU - no torch.cuda call, no libsmbv.so load - and never replaces itself: a process that has initialised HIP must not exec (and need not: it is a child)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")         # RCCL's kernels on high-priority streams: not queued behind 2-ms attention blocks (dp.py)
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_threads() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    log("[bench] self-launch:", " ".join(cmd))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    lines = 0
    for out in proc.stdout:
        if out.lstrip().startswith('{"metric"'):
            sys.stdout.write(out)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        log(f"[bench] the ranks exited 0 but printed {lines} JSON lines")
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-roofline", action="store_true")
    ap.add_argument("--no-feed", action="store_true", help="skip the second timed region (new volume + mask per step through the VolumeFeeder)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip config.vjepa_ms_per_step / config.vit_large_fp8_recompute_ms_per_step / config.reference_class_on_gpu")
    ap.add_argument("--no-reference-class", action="store_true", help="skip cpu_baseline.reference_class (transformers.VideoMAEForPreTraining on the host cores)")
    ap.add_argument("--no-trainer", action="store_true", help="skip config.trainer_ms_per_step (the HF-Trainer harness at full size)")
    ap.add_argument("--no-optimizer", action="store_true", help="time forward+backward only (diagnostic; not the metric)")
    ap.add_argument("--torch-optimizer", action="store_true", help="clip_grad_norm_ + torch fused AdamW instead of the fused HIP optimizer tail")
    ap.add_argument("--tiny", action="store_true", help="64x64x32 smoke shape (diagnostic; not the metric)")
    ap.add_argument("--vjepa", action="store_true", help="BASELINE.json config 4: the V-JEPA train step (diagnostic line; not the metric)")
    ap.add_argument("--vit-large", action="store_true", help="BASELINE.json config 5's model in bf16 (encoder 1024/24L/16H/4096; decoder 512/12L/8H/2048 - the reference "
                                                               "specifies none, SURVEY.md 8d) - diagnostic; not the metric")
    ap.add_argument("--fp8", action="store_true", help="attention forward on the block-scaled fp8 MFMA (config.smbv_attention_fp8; BASELINE config 5 with --vit-large) - diagnostic")
    ap.add_argument("--recompute", action="store_true", help="gradient checkpointing: per-layer recompute (config.smbv_recompute = always) - diagnostic")
    ap.add_argument("--debug-variant", type=int, default=0, help="smbv_debug_set_variant bits (kernel A/B; diagnostic, not the metric)")
    ap.add_argument("--dp-grad-dtype", default="fp32", choices=("fp32", "bf16"), help="what the N>1 gradient all-reduce ships (bf16 = DDP's bf16_compress_hook; diagnostic)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only to rehearse the N>1 path on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: this process makes NO GPU call and starts the N ranks itself, as children
        raise SystemExit(self_launch(args.gpus))
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it with --nproc-per-node {args.gpus} (or with no launcher at all)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MIM hot path has no CPU fallback")
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    if args.debug_variant:
        from smb_vision_amd import _lib
        _lib.load().smbv_debug_set_variant(args.debug_variant)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    if args.vjepa:
        run_vjepa(args, device, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    cfg_kw = dict(VIT_BASE)
    if args.tiny:
        cfg_kw.update(image_size=64, num_frames=32, hidden_size=128, num_attention_heads=2, num_hidden_layers=2, intermediate_size=512,
                      decoder_hidden_size=64, decoder_num_attention_heads=1, decoder_num_hidden_layers=1, decoder_intermediate_size=256)
    if args.vit_large:
        cfg_kw.update(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, decoder_hidden_size=512,
                      decoder_num_hidden_layers=12, decoder_num_attention_heads=8, decoder_intermediate_size=2048)
    if args.fp8:
        cfg_kw["smbv_attention_fp8"] = "all"
    if args.recompute:
        cfg_kw["smbv_recompute"] = "always"
    model, pix, mask, gen = build(rank, device, cfg_kw)
    if args.recompute:
        model.gradient_checkpointing_enable()
    n_masked = gen.num_masked
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
        dp = DataParallelEngine(model, grad_dtype=args.dp_grad_dtype)
        dp.broadcast_parameters(model)

    def train(pixel_values, **mask_kw):
        out = model(pixel_values=pixel_values, **mask_kw)
        out.loss.backward()
        if opt is not None:
            if args.torch_optimizer:
                torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
            opt.step()
            opt.zero_grad(set_to_none=True)
        else:
            model.zero_grad(set_to_none=True)
        return out.loss

    last = {}
    from smb_vision_amd.masking import split_mask_indices_host
    vis_h, msk_h = split_mask_indices_host(mask.cpu())      # what run_mim.collate_fn hands over next to the mask (no per-step device sort)
    vis_d, msk_d = vis_h.to(device), msk_h.to(device)

    def step():
        last["loss"] = train(pix, bool_masked_pos=mask, visible_index=vis_d, masked_index=msk_d)

    for _ in range(args.warmup):
        step()
    dt = timed_region(step, args.steps, world, device)
    final_loss = float(last["loss"].detach())

    # second timed region: SURVEY.md §8(d)'s step - a NEW volume from host memory and a NEW mask every step
    fed = None
    if not args.no_feed and not args.tiny:  # (N > 1: every rank feeds its own volumes - PCIe contention shows in the scaling runs)
        from smb_vision_amd.feed import VolumeFeeder
        feeder = VolumeFeeder(_FedSource(rank, gen, cfg_kw["image_size"], cfg_kw["num_frames"]), device, cfg_kw["image_size"], cfg_kw["num_frames"])
        it = iter(feeder)

        def fed_step():
            p, ex = next(it)
            last["loss"] = train(p, bool_masked_pos=ex["mask"].to(device, non_blocking=True), visible_index=ex["vis"], masked_index=ex["msk"])

        # warm-up covers every page-locked source buffer once (4 volumes + the one in flight): the FIRST upload out of a freshly pinned region runs at a
        # fraction of the PCIe rate (steps 1-6 from a cold start take 30-34 ms, tools/fed_breakdown.py history in DESIGN.md §5); a training run
        # recycles its pinned buffers, so the steady state is what the figure should show
        # (and the caching host allocator keeps adding small page-locked blocks for the per-step mask / index tensors until about step 14)
        FED_WARMUP = max(16, args.warmup)
        for _ in range(FED_WARMUP):
            fed_step()
        dt_fed = timed_region(fed_step, args.steps, world, device)
        it.close()
        fed = {"ms_per_step": dt_fed / args.steps * 1e3, "value": world * args.steps / dt_fed, "unit": "volumes/s", "final_loss": float(last["loss"].detach()),
               "what": "every step: new int16 volume (167.8 MB) from pinned host memory via VolumeFeeder (copy stream, device-side scale/pad/crop/permute) + new mask, "
                       f"host-built token lists; same barrier protocol; {FED_WARMUP} warm-up steps (first uploads out of freshly page-locked buffers are slow)"}

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
            "vs_baseline": None, "dtype": "bf16" if not args.fp8 else "bf16 (attention forward fp8 e4m3)", "data": "synthetic",
            "config": {"workload": "ViT-base MIM, 512x512x320 volume, patch_size=16, mask_ratio=0.65 (mask_patch 32), B=1/GPU, "
                                   "fwd+loss+bwd+clip+AdamW, volume resident in HBM" + ("" if opt is not None else " [NO OPTIMIZER: diagnostic]")
                                   + (" [TINY: diagnostic]" if args.tiny else "") + (" [ViT-LARGE: diagnostic, 64.41 TFLOP/volume]" if args.vit_large else "")
                                   + (" [fp8 e4m3 attention forward: diagnostic]" if args.fp8 else "") + (" [per-layer recompute]" if args.recompute else ""),
                       "tokens": model.num_patches, "masked": n_masked, "parallelism": f"dp{world}", "final_loss": final_loss,
                       "resident_ms_per_step": ms, "replicas_in_sync": in_sync},
        }
        if dp is not None and dp._buckets:   # what travelled: the all-reduce plan of the last step (dp.py) and whether the host had every bucket's collective queued before its gradients existed
            line["config"]["dp"] = {"backend": args.backend, "grad_dtype": dp.grad_dtype, "bucket_MB": [round((b - a) * 4 / 1e6, 1) for a, b, _ in dp._buckets],
                                    "overlap": dp.comm_stream is not None, "enqueued_before_ready": [bool(x) for x in dp.enqueue_lead]}
        if fed is not None:
            line["fed"] = fed
        step_flop = 64.41 if args.vit_large else STEP_TFLOP
        step_tf = world * step_flop / (ms * 1e-3) if not args.tiny else None  # whole job, TFLOP/s
        if world > 1 and not args.tiny:  # multi-GPU lines carry the whole-step roofline only (the kernel figure is a 1-GPU property)
            line["roofline"] = {"bound": "mfma", "kernel": "whole step (per GPU)", "achieved": step_tf / world, "peak": MFMA_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": step_tf / world / MFMA_PEAK_TFLOPS, "traffic": None}
        elif not args.no_kernel_roofline and not args.tiny:
            kr, unit = kernel_roofline(device)
            traffic = None
            try:  # PMC-measured fabric bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes), committed under profiles/
                traffic = json.load(open(os.path.join(ROOT, "profiles", "attn_bwd_traffic.json")))["traffic_bytes_per_launch"]
            except Exception:
                pass
            pair_s = kr["pair"] * 1e-3
            # ONE convention for the line and for the step figure: SURVEY.md 8(d) credits attention backward at 2 x forward = FOUR S x S x Dh products per
            # launch (8 S^2 Dh H = 1.2885 TFLOP) - that is `achieved` / `frac`.  Beside it: the five products a fused flash backward executes
            # (`frac_5_products`: S, dP, dV, dK, dQ) and the seven this launch executes (`executed_frac`: its query-owner blocks recompute S and dP).
            line["roofline"] = {"bound": "mfma", "kernel": "decoder attention backward: attn_bwd_pair_kernel - key-owner (dK/dV) and query-owner (dQ) blocks mixed per CU in one launch (S=20480, H=6, Dh=64)",
                                "achieved": 4 * unit / pair_s / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": 4 * unit / pair_s / 1e12 / MFMA_PEAK_TFLOPS,
                                "frac_5_products": 5 * unit / pair_s / 1e12 / MFMA_PEAK_TFLOPS,
                                "executed_frac": 7 * unit / pair_s / 1e12 / MFMA_PEAK_TFLOPS, "traffic": traffic, "launch_ms": kr["pair"],
                                "flop_per_launch": 4 * unit, "flop_per_launch_5_products": 5 * unit, "executed_flop_per_launch": 7 * unit,
                                "dkv": {"launch_ms": kr["dkv"], "achieved": 4 * unit / (kr["dkv"] * 1e-3) / 1e12, "frac": 4 * unit / (kr["dkv"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                        "flop_per_launch": 4 * unit},
                                "dq": {"launch_ms": kr["dq"], "executed": 3 * unit / (kr["dq"] * 1e-3) / 1e12},
                                "step": {"achieved": step_tf, "flop_per_volume_T": step_flop, "frac": step_tf / MFMA_PEAK_TFLOPS}}
        if world == 1 and not args.no_trainer and not args.tiny and not args.vit_large and not args.fp8 and not args.recompute:
            try:
                line["config"]["trainer_ms_per_step"] = trainer_ms_per_step(cfg_kw)
            except Exception as e:
                line["config"]["trainer_ms_per_step"] = None
                line["config"]["trainer_error"] = repr(e)
        plain = not args.tiny and not args.vit_large and not args.fp8 and not args.recompute and opt is not None and not args.torch_optimizer
        if world == 1 and plain and not args.no_extra_configs:
            # SURVEY.md 8(d) "with and without optimizer": forward + loss + backward only, same model and volume, same barrier protocol
            try:
                def step_noopt():
                    model(pixel_values=pix, bool_masked_pos=mask, visible_index=vis_d, masked_index=msk_d).loss.backward()
                    model.zero_grad(set_to_none=True)
                for _ in range(2):
                    step_noopt()
                line["config"]["no_optimizer_ms_per_step"] = timed_region(step_noopt, args.steps, 1, device) / args.steps * 1e3
            except Exception as e:
                line["config"]["no_optimizer_ms_per_step"], line["config"]["no_optimizer_error"] = None, repr(e)
            # the reference's SHIPPED recipe runs four volumes per device (scripts/training/run_mim.sh:13-38: --per_device_train_batch_size 4); the
            # headline stays README.md:56-68's B = 1
            try:
                line["config"]["b4_ms_per_volume"], line["config"]["b4"] = b4_leg(model, opt, gen, pix, device)
            except Exception as e:
                line["config"]["b4_ms_per_volume"], line["config"]["b4_error"] = None, repr(e)
        if world == 1 and plain and not args.no_extra_configs:  # BASELINE.json configs 4 and 5 on one GPU, driver-visible
            try:
                dtv, info = vjepa_leg(device, 1, 0, steps=5, warmup=2)
                line["config"]["vjepa_ms_per_step"] = dtv / 5 * 1e3
                line["config"]["vjepa"] = dict(info, tflop_per_step=VJEPA_STEP_TFLOP, frac_of_mfma_peak=VJEPA_STEP_TFLOP / (dtv / 5) / MFMA_PEAK_TFLOPS,
                                               what="BASELINE.json config 4: V-JEPA ViT-base encoder + predictor, 512x512x320, B=1, 3 mask blocks")
            except Exception as e:
                line["config"]["vjepa_ms_per_step"], line["config"]["vjepa_error"] = None, repr(e)
            try:
                msl = vit_large_leg(device)
                line["config"]["vit_large_fp8_recompute_ms_per_step"] = msl
                line["config"]["vit_large"] = dict(tflop_per_step=64.41, frac_of_mfma_peak_excl_recompute=64.41 / (msl * 1e-3) / MFMA_PEAK_TFLOPS,
                                                   what="BASELINE.json config 5's model on one GPU: ViT-large MIM, fp8 e4m3 attention forward + per-layer recompute")
            except Exception as e:
                line["config"]["vit_large_fp8_recompute_ms_per_step"], line["config"]["vit_large_error"] = None, repr(e)
            try:
                line["config"]["reference_class_on_gpu"] = reference_class_on_gpu(device)
            except Exception as e:
                line["config"]["reference_class_on_gpu"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not args.tiny:
            try:
                line["cpu_baseline"] = cpu_baseline(cpu_threads())
            except Exception as e:  # the baseline leg must never take the bench line down
                line["cpu_baseline"] = {"error": repr(e)}
            if "error" not in line["cpu_baseline"] and not args.no_reference_class:
                try:
                    line["cpu_baseline"]["reference_class"] = reference_class_baseline(cpu_threads())
                except Exception as e:
                    line["cpu_baseline"]["reference_class"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
