"""A/B of attention kernel variants (smbv_debug_set_variant bits) in ONE process, interleaved rounds (MI355X guide rule 24): per shape and kernel the
median / min over the rounds, and a bitwise comparison of the variants' results.  usage: attn_ab.py [variant_bits ...]  (default: 0 and 1<<24, the
XCD-aware block order)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
lib = _lib.load()
variants = [int(a) for a in sys.argv[1:]] or [0, 1 << 24]
ROUNDS, ITERS = 7, 6
def t_of(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for S, H, Dh, Sq, qoff in ((20480, 6, 64, 0, 0), (7168, 12, 64, 0, 0), (20480, 6, 64, 13312, 7168), (20480, 12, 32, 0, 0), (20480, 12, 64, 0, 0)):
    w = H * Dh
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H, Dh=Dh, Sq=Sq, q_off=qoff)
    out = (torch.zeros_like(qkv), torch.empty(1, H, Sq or S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out, Dh=Dh, Sq=Sq, q_off=qoff)
    kinds = {"fwd": lambda: ops.attention_fwd(qkv, 1, S, H, Dh=Dh, Sq=Sq, q_off=qoff),
             "dkv": lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=2, out=out, Dh=Dh, Sq=Sq, q_off=qoff),
             "dq": lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=4, out=out, Dh=Dh, Sq=Sq, q_off=qoff),
             "pair": lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out, Dh=Dh, Sq=Sq, q_off=qoff)}
    # results must not depend on the block order
    ref = None
    for v in variants:
        lib.smbv_debug_set_variant(v)
        Ov, lv = ops.attention_fwd(qkv, 1, S, H, Dh=Dh, Sq=Sq, q_off=qoff)
        o2 = (torch.zeros_like(qkv), torch.empty(1, H, Sq or S, dtype=torch.float32, device="cuda"))
        ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=o2, Dh=Dh, Sq=Sq, q_off=qoff)
        cur = (Ov.clone(), lv.clone(), o2[0].clone())
        if ref is None: ref = cur
        elif (v ^ variants[0]) & 3: pass   # bits 0 / 1: the key-owner / query-owner block on the other MFMA shape (another summation order)
        else: assert all(torch.equal(a, b) for a, b in zip(ref, cur)), f"variant {v} changes the results"
    times = {(k, v): [] for k in kinds for v in variants}
    for _ in range(2):
        for k, fn in kinds.items(): fn()
    for r in range(ROUNDS):
        for k, fn in kinds.items():
            for v in variants:
                lib.smbv_debug_set_variant(v)
                times[(k, v)].append(t_of(fn))
    lib.smbv_debug_set_variant(0)
    print(f"S{S} H{H} Dh{Dh} Sq{Sq}:")
    for k in kinds:
        print("   " + f"{k:5s}" + " | ".join(f" v{v}: median {statistics.median(times[(k, v)]):8.1f} us  min {min(times[(k, v)]):8.1f}" for v in variants), flush=True)
