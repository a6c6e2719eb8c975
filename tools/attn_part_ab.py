"""A/B one part of the attention backward (parts: 2 = dK/dV kernel, 4 = dQ kernel, 6 = both forked) across tuning variants.
usage: attn_part_ab.py PARTS V0,V1[,V2...]   (variants are smbv_debug_set_variant bit masks; outputs compared bitwise against the first)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1024]
def timeit(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
nprod = {2: 8, 4: 6, 6: 14}[parts]
for name, S, H, Sq, qoff in (("enc", 7168, 12, 0, 0), ("dec", 20480, 6, 0, 0), ("dec-tail", 20480, 6, 13312, 7168), ("ragged", 1000, 3, 0, 0)):
    w = H * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H, Sq=Sq, q_off=qoff) if Sq else ops.attention_fwd(qkv, 1, S, H)
    sq = Sq if Sq else S
    delta = torch.empty(1, H, sq, dtype=torch.float32, device="cuda")
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=1, out=(torch.zeros_like(qkv), delta), Sq=Sq, q_off=qoff)
    outs, res = {}, {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            raw.smbv_debug_set_variant(v)
            out = (torch.zeros_like(qkv), delta)
            ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=parts, out=out, Sq=Sq, q_off=qoff); torch.cuda.synchronize()
            outs[v] = out[0].clone()
            res[v].append(timeit(lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=parts, out=out, Sq=Sq, q_off=qoff)))
    fl = nprod * S * sq * 64 * H
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
    print(f"{name:9s} parts={parts}: " + "  ".join(f"v{v}: {sorted(res[v])[2]*1e3:8.1f}us {fl/sorted(res[v])[2]/1e9:6.1f}TF" for v in variants) + f"  same={same}", flush=True)
raw.smbv_debug_set_variant(0)
