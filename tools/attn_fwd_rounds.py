"""How much does the partial last round of the attention forward cost?  attn_fwd3_kernel holds 3 blocks of 128 queries per CU (768 slots); the grid is
(S / 128) * H blocks.  Same S, varying H: time per block-round tells what a grid of 1.25 rounds (the MIM decoder: 960 blocks) loses against whole rounds.
usage: python tools/attn_fwd_rounds.py [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
heads = (3, 4, 6, 8, 9, 12, 14, 24)
bufs = {H: torch.randn(S, 3 * H * 64, generator=g, device="cuda").bfloat16() for H in heads}
for rnd in range(3):
    for H in heads:
        qkv = bufs[H]
        for _ in range(3):
            ops.attention_fwd(qkv, 1, S, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention_fwd(qkv, 1, S, H)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(H, []).append(e0.elapsed_time(e1) / 10)
for H in heads:
    t = sorted(res[H])[1]
    blocks = (S + 127) // 128 * H
    print(f"S {S} H {H:2d}: {blocks:5d} blocks = {blocks / 768:5.2f} rounds of 768   {t * 1e3:7.1f} us   {4.0 * S * S * 64 * H / t / 1e9:6.0f} TFLOP/s   {t * 1e3 / (blocks / 768):7.1f} us per round-equivalent")
