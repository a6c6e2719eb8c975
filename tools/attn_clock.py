"""How much of the attention kernels' time is the clock the chip holds under load (MI355X_MICROARCH.md 'DVFS give-back' item 1): the same launches on random
and on all-zero operands (same instruction stream - neither kernel branches on data here), back to back for >= 1 s each.  usage: attn_clock.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
def t_of(fn, seconds=1.0):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / n
    n = max(20, int(seconds * 1e3 / per))
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for S, H, Dh in ((20480, 6, 64), (7168, 12, 64), (20480, 12, 32)):
    w = H * Dh
    res = {}
    for kind in ("random", "zeros", "random"):
        if kind == "random":
            qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
            dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
        else:
            qkv = torch.zeros(S, 3 * w, device="cuda").bfloat16()
            dO = torch.zeros(S, w, device="cuda").bfloat16()
        O, lse = ops.attention_fwd(qkv, 1, S, H, Dh=Dh)
        out = (torch.zeros_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
        ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out, Dh=Dh)
        tf = t_of(lambda: ops.attention_fwd(qkv, 1, S, H, Dh=Dh))
        tb = t_of(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out, Dh=Dh))
        print(f"S{S} H{H} Dh{Dh} {kind:7s}: forward {tf:8.1f} us   backward pair {tb:8.1f} us", flush=True)
