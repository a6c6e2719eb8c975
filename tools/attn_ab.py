import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
def timeit(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 2, 4, 8]
for name, S, H in (("enc", 7168, 12), ("dec", 20480, 6)):
    w = H * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H)
    out = (torch.empty_like(qkv), torch.empty(1, H, S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=1, out=out)
    f = 4.0 * S * S * 64 * H
    kern = {"fwd": (lambda: ops.attention_fwd(qkv, 1, S, H), f), "dkv": (lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=2, out=out), 2 * f),
            "dq": (lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=4, out=out), 1.5 * f)}
    for kname, (fn, flop) in kern.items():
        res = {v: [] for v in variants}
        for rnd in range(5):
            for v in variants:
                raw.smbv_set_variant(v)
                fn(); torch.cuda.synchronize()
                res[v].append(timeit(fn))
        print(f"{name} {kname}: " + "  ".join(f"v{v}: {sorted(res[v])[2]*1e3:7.1f}us {flop/sorted(res[v])[2]/1e9:6.1f}TF" for v in variants))
raw.smbv_set_variant(0)
