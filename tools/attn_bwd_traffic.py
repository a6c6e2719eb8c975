"""Target of the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for bench.py's roofline.traffic: the decoder attention backward (dK/dV kernel and dQ kernel,
S = 20,480, 6 heads) run a few times, one kernel at a time (parts 2, then 4) so that every dispatch is attributed to its kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
S, H = 20480, 6
w = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
O, lse = ops.attention_fwd(qkv, 1, S, H)
out = (torch.empty_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=1, out=out)
for _ in range(3):
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=2, out=out)
    torch.cuda.synchronize()
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=4, out=out)
    torch.cuda.synchronize()
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out)   # the shipped form: one launch, mixed block roles (attn_bwd_pair_kernel)
    torch.cuda.synchronize()
