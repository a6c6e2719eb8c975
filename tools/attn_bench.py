"""Attention kernels at the models' shapes: forward, dK/dV, dQ and the concurrent backward pair, per head width (64: MIM / V-JEPA encoder,
32: V-JEPA predictor).  FLOP counted as 4 S^2 Dh H per product pair (forward = 1 pair, dK/dV = 2, dQ = 1.5, algorithmic backward = 2.5)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
def timeit(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device="cuda").manual_seed(0)
for S, H, Dh in ((20480, 6, 64), (7168, 12, 64), (20480, 12, 32), (8258, 12, 64), (20480, 12, 64)):
    w = H * Dh
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H, Dh=Dh)
    out = (torch.zeros_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out, Dh=Dh)
    unit = 4.0 * S * S * Dh * H
    tf = timeit(lambda: ops.attention_fwd(qkv, 1, S, H, Dh=Dh))
    tkv = timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=2, out=out, Dh=Dh))
    tq = timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=4, out=out, Dh=Dh))
    tp = timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out, Dh=Dh))
    print(f"S{S} H{H} Dh{Dh}: fwd {tf*1e3:7.1f} us {unit/tf/1e9:5.0f} TF | dkv {tkv*1e3:7.1f} us {2*unit/tkv/1e9:5.0f} TF | dq {tq*1e3:7.1f} us {1.5*unit/tq/1e9:5.0f} TF | "
          f"pair {tp*1e3:7.1f} us {2.5*unit/tp/1e9:5.0f} TF (5 products) {3.5*unit/tp/1e9:5.0f} TF (7 executed)", flush=True)
