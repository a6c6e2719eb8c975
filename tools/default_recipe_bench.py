"""The reference's SHIPPED recipe at its argument defaults (scripts/training/run_mim.sh + src/run_mim.py DataTrainingArguments): ViT-base, 224 x 224 x 160
volumes (1,960 tokens), mask patches of 32, mask ratio 0.5, 4 volumes per step, bf16, gradient checkpointing on.  A much smaller problem than the
north-star shape (20,480 tokens): kernels of 10-100 us, so what matters is whether the HOST can enqueue a step as fast as the GPU runs it.
Prints GPU ms/step (events over K steps), host ms/step (time for the Python step to return, no sync) and volumes/s, with and without recompute.
usage: python tools/default_recipe_bench.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from transformers import VideoMAEConfig
from smb_vision_amd.masking import MaskGenerator, split_mask_indices_host
from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
from smb_vision_amd.optim import SmbvAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
for recompute in (True, False):
    kw = dict(num_channels=1, image_size=224, num_frames=160, patch_size=16, tubelet_size=16)
    if recompute:
        kw["smbv_recompute"] = "always"
    torch.manual_seed(0)
    model = SmbVideoMAEForPreTraining(VideoMAEConfig(**kw)).to(dev).train()
    if recompute:
        model.gradient_checkpointing_enable()
    opt = SmbvAdamW(list(model.parameters()), lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
    np.random.seed(0)
    gen = MaskGenerator(224, 160, 32, 16, 0.5)
    mask = torch.stack([gen() for _ in range(B)])
    vis, msk = split_mask_indices_host(mask)
    mask, vis, msk = mask.to(dev), vis.to(dev), msk.to(dev)
    pix = torch.rand(B, 160, 1, 224, 224, device=dev)

    def step():
        out = model(pixel_values=pix, bool_masked_pos=mask, visible_index=vis, masked_index=msk)
        out.loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return out.loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = 0.0
    e0.record()
    for _ in range(K):
        t = time.perf_counter()
        loss = step()
        host += time.perf_counter() - t
    e1.record()
    torch.cuda.synchronize()
    gpu = e0.elapsed_time(e1) / K
    print(f"B {B} recompute {recompute}: {gpu:.2f} ms/step on the GPU timeline, host enqueue {host / K * 1e3:.2f} ms/step -> {B / gpu * 1e3:.0f} volumes/s "
          f"(loss {float(loss):.4f}; tokens {gen.num_masked} masked of 1960)", flush=True)
    del model, opt
    torch.cuda.empty_cache()
