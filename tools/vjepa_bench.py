"""Diagnostic timing of the V-JEPA train step (BASELINE.json config 4: ViT-base encoder + default predictor, 512x512x320, B=1):
online forward + frozen target-encoder forward + L1 + EMA + backward + fused clip/AdamW.  Not the headline metric."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import VJEPA2Config
from smb_vision_amd.modeling_vjepa import MomentumEncoder, SmbVJEPA2Model, make_target_encoder, vjepa_loss
from smb_vision_amd.optim import SmbvAdamW

small = "--small" in sys.argv
cfg = VJEPA2Config(patch_size=16, crop_size=128 if small else 512, frames_per_clip=64 if small else 320, tubelet_size=16, in_chans=1, hidden_size=768,
                   num_hidden_layers=12, num_attention_heads=12, mlp_ratio=4.0, pred_hidden_size=384, pred_num_hidden_layers=12,
                   pred_num_attention_heads=12)
torch.manual_seed(0)
model = SmbVJEPA2Model(cfg).cuda().train()
target = make_target_encoder(model)
ema = MomentumEncoder(target, 0.99925)
opt = SmbvAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, max_grad_norm=1.0)
N = model.num_patches
g = torch.Generator(device="cuda").manual_seed(1)
pix = torch.rand(1, cfg.frames_per_clip, 1, cfg.crop_size, cfg.crop_size, generator=g, device="cuda")
perm = torch.randperm(N, generator=torch.Generator().manual_seed(2))
n_tgt = int(0.5 * N)
tm = perm[:n_tgt].sort().values.unsqueeze(0).cuda()
cm = perm[n_tgt:].sort().values.unsqueeze(0).cuda()
inputs = {"pixel_values_videos": pix, "context_mask": [cm], "target_mask": [tm]}

def step():
    loss, _ = vjepa_loss(model, target, ema, inputs)
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return loss

for _ in range(2):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flops = 102.2e12 if not small else None
print(f"V-JEPA step: {dt*1e3:.1f} ms  ({1/dt:.2f} volumes/s)  tokens {N} ctx {cm.shape[1]} tgt {tm.shape[1]}  loss {float(loss):.4f}" +
      (f"  ~{flops/dt/1e12:.0f} TFLOP/s of the nominal 102.2 TFLOP step" if flops else ""))
