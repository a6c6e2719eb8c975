import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
def timeit(fn, iters=5):
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
    outs = {}
    res = {0: [], 8192: []}
    for rnd in range(5):
        for v in res:
            raw.smbv_set_variant(v)
            out = (torch.zeros_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
            ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out); torch.cuda.synchronize()
            outs[v] = out[0].clone()
            res[v].append(timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out)))
    print(f"{name} bwd (delta+dkv+dq): concurrent {sorted(res[0])[2]*1e3:8.1f} us   serial {sorted(res[8192])[2]*1e3:8.1f} us  same={torch.equal(outs[0], outs[8192])}")
raw.smbv_set_variant(0)
