import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
S, H = 20480, 6
w = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
t = timeit(lambda: ops.attention_fwd(qkv, 1, S, H))
print(f"VAR={os.environ.get('SMBV_ATTN_VAR','0')} dec fwd {t*1e3:8.1f} us")
