"""A/B the attention forward across tuning variants (smbv_debug_set_variant masks): timing + max deviation of O / LSE from the first variant
and from a float64 torch reference on the small shape.  usage: attn_fwd_ab.py V0,V1,..."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 32]
def timeit(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, S, H, Sq, qoff in (("enc", 7168, 12, 0, 0), ("dec", 20480, 6, 0, 0), ("dec-tail", 20480, 6, 13312, 7168), ("ragged", 1000, 3, 0, 0), ("tiny", 70, 2, 0, 0), ("spiky", 2048, 2, 0, 0)):
    w = H * 64
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda")
    if name == "spiky": qkv[:, :2 * w] *= 6.0   # large logits: forces rebases along the key axis
    qkv = qkv.bfloat16()
    outs, res = {}, {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            raw.smbv_debug_set_variant(v)
            O, lse = ops.attention_fwd(qkv, 1, S, H, Sq=Sq, q_off=qoff); torch.cuda.synchronize()
            outs[v] = (O.float().clone(), lse.clone())
            res[v].append(timeit(lambda: ops.attention_fwd(qkv, 1, S, H, Sq=Sq, q_off=qoff)))
    sq = Sq if Sq else S
    fl = 4.0 * S * sq * 64 * H
    ref = outs[variants[0]]
    dev = "  ".join(f"v{v}: dO={float((outs[v][0]-ref[0]).abs().max()):.2e} dLSE={float((outs[v][1]-ref[1]).abs().max()):.2e}" for v in variants[1:])
    extra = ""
    if S <= 2048:
        q, k, vv = (qkv[:, i * w:(i + 1) * w].double().view(S, H, 64).transpose(0, 1) for i in range(3))
        sc = q @ k.transpose(1, 2) * 0.125
        Or = (torch.softmax(sc, -1) @ vv).transpose(0, 1).reshape(S, w)
        Lr = torch.logsumexp(sc, -1)
        extra = "  vs f64: " + "  ".join(f"v{v}: dO={float((outs[v][0].double()-Or).abs().max()):.2e} dLSE={float((outs[v][1][0].double()-Lr).abs().max()):.2e}" for v in variants)
    print(f"{name:9s}: " + "  ".join(f"v{v}: {sorted(res[v])[2]*1e3:8.1f}us {fl/sorted(res[v])[2]/1e9:6.1f}TF" for v in variants) + "  " + dev + extra, flush=True)
raw.smbv_debug_set_variant(0)
