"""The attention backward pair as ONE launch with per-CU mixed block roles (variant 0, the default) against two launches on two streams (variant 524288): interleaved
rounds in one process, bitwise comparison of the gradients, at the step's shapes.  usage: attn_pair_ab.py [variant bits ...]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
lib = _lib.load()
variants = [int(a) for a in sys.argv[1:]] or [0, 524288]
ROUNDS, ITERS = 9, 6
def t_of(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for S, H, Dh, Sq, qoff in ((20480, 6, 64, 0, 0), (7168, 12, 64, 0, 0), (20480, 6, 64, 13312, 7168), (20480, 12, 32, 0, 0), (20480, 12, 64, 0, 0), (1000, 3, 64, 0, 0), (777, 2, 32, 300, 100)):
    w = H * Dh
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, 1, S, H, Dh=Dh, Sq=Sq, q_off=qoff)
    out = (torch.zeros_like(qkv), torch.empty(1, H, Sq or S, dtype=torch.float32, device="cuda"))
    ops.attention_bwd(qkv, O, dO, lse, 1, S, H, out=out, Dh=Dh, Sq=Sq, q_off=qoff)   # delta
    ref = None
    for v in variants:
        lib.smbv_debug_set_variant(v)
        o2 = (torch.zeros_like(qkv), out[1])
        for _ in range(3):
            ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=o2, Dh=Dh, Sq=Sq, q_off=qoff)
        torch.cuda.synchronize()
        cur = o2[0].clone()
        if ref is None: ref, vref = cur, v
        elif (v ^ vref) & 3:   # bits 0 / 1 pick the other MFMA shape for the key-owner / query-owner block: another summation order, not bitwise the same
            print(f"  v{v} against v{vref}: max |diff| / max |ref| = {float((cur.float() - ref.float()).abs().max() / ref.float().abs().max()):.2e}")
        else: assert torch.equal(ref, cur), f"variant {v} changes the gradients at S{S} H{H} Dh{Dh} Sq{Sq}"
    times = {v: [] for v in variants}
    fn = lambda: ops.attention_bwd(qkv, O, dO, lse, 1, S, H, parts=6, out=out, Dh=Dh, Sq=Sq, q_off=qoff)
    for r in range(ROUNDS):
        for v in variants:
            lib.smbv_debug_set_variant(v)
            fn(); times[v].append(t_of(fn))
    lib.smbv_debug_set_variant(0)
    print(f"S{S} H{H} Dh{Dh} Sq{Sq}: " + " | ".join(f"v{v}: median {statistics.median(times[v]):8.1f} us  min {min(times[v]):8.1f}" for v in variants), flush=True)
