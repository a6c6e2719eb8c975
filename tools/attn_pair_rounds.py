"""How much does the partial last round of the attention backward pair launch cost?  attn_bwd_pair_kernel holds 2 blocks per CU (512 slots); the grid is
2 * (S / 128) * H blocks (key-owner + query-owner tiles), claimed from two queues.  Same H, varying S: time / S^2 tells what 2.6 rounds (the MIM encoder) lose
against whole rounds.   usage: python tools/attn_pair_rounds.py [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 12
g = torch.Generator(device="cuda").manual_seed(0)
sizes = (5120, 6144, 7168, 8192, 9216, 10240, 12288) if H == 12 else (10240, 16384, 20480, 21760, 24576)
data = {}
for S in sizes:
    w = H * 64
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H)
    out = (torch.empty_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=1, out=out)
    data[S] = (qkv, O, dO, lse, out)
res = {}
for rnd in range(3):
    for S in sizes:
        qkv, O, dO, lse, out = data[S]
        for _ in range(3):
            ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(S, []).append(e0.elapsed_time(e1) / 10)
for S in sizes:
    t = sorted(res[S])[1]
    blocks = 2 * ((S + 127) // 128) * H
    print(f"H {H} S {S:6d}: {blocks:5d} tiles = {blocks / 512:5.2f} rounds of 512   {t * 1e3:7.1f} us   executed {14.0 * S * S * 64 * H / t / 1e9:6.0f} TFLOP/s   {t * 1e9 / (S * S):7.3f} ps per (query, key) pair")
