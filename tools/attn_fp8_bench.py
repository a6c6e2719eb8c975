"""Attention forward, bf16 kernel vs the fp8 (block-scaled MFMA) path incl. its quantisation pre-passes, at the model's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
def timeit(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device="cuda").manual_seed(0)
for S, H in ((20480, 6), (7168, 12), (20480, 8), (7168, 16)):
    qkv = torch.randn(S, 3 * H * 64, generator=g, device="cuda").bfloat16()
    tb = timeit(lambda: ops.attention_fwd(qkv, 1, S, H))
    tf = timeit(lambda: ops.attention_fwd_fp8(qkv, 1, S, H))
    fl = 4.0 * S * S * 64 * H
    print(f"S{S} H{H}: bf16 {tb*1e3:7.1f} us {fl/tb/1e9:6.0f} TF | fp8 (with quantisation) {tf*1e3:7.1f} us {fl/tf/1e9:6.0f} TF", flush=True)
