"""Embedding export rate (src/run_inference.py:78-83: model.videomae(image), all 20,480 tokens of a 512x512x320 volume, ViT-base).
19.07 TFLOP per volume (encoder forward over all tokens)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import VideoMAEConfig
from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEModel

cfg = VideoMAEConfig(image_size=512, num_frames=320, patch_size=16, tubelet_size=16, num_channels=1, hidden_size=768, num_hidden_layers=12,
                     num_attention_heads=12, intermediate_size=3072, use_mean_pooling=True)
torch.manual_seed(0)
m = SmbVideoMAEModel(cfg).cuda().eval()
pix = torch.rand(1, 320, 1, 512, 512, device="cuda")
with torch.no_grad():
    for _ in range(2):
        out = m(pix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        out = m(pix)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
print(f"embedding export: {dt*1e3:.1f} ms/volume  ({1/dt:.1f} volumes/s, {19.07/dt:.0f} TFLOP/s)  out {tuple(out.last_hidden_state.shape)} finite {bool(torch.isfinite(out.last_hidden_state).all())}")
