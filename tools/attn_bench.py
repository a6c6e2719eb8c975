"""Micro-benchmark of the attention kernels at the bench shapes (encoder S=7168/H=12, decoder S=20480/H=6)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, S, H in (("enc", 7168, 12), ("dec", 20480, 6)):
    w = H * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H)
    out = (torch.empty_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=1, out=out)
    f = 4.0 * S * S * 64 * H
    t = timeit(lambda: ops.attention_fwd(qkv, 1, S, H)); print(f"{name} fwd  {t*1e3:8.1f} us  {f/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=2, out=out)); print(f"{name} dkv  {t*1e3:8.1f} us  {2*f/t/1e9:7.1f} TF/s")
    t = timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=4, out=out)); print(f"{name} dq   {t*1e3:8.1f} us  {1.5*f/t/1e9:7.1f} TF/s")
