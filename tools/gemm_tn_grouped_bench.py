"""A transformer block's four weight-gradient products: ONE grouped launch (smbv_gemm_tn_grouped: every tile over all tokens, no slabs) against the four
per-product launches (smbv_gemm_tn_bias: token slabs + finishing kernel), alone on the chip.  usage: gemm_tn_grouped_bench.py [variant bits]
(32768 = the grouped launch on 128 x 128 tiles where 256 x 256 would be taken)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib, ops

_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for name, M, w, ff in [("ViT-base encoder", 7168, 768, 3072), ("ViT-base decoder", 20480, 384, 1536), ("ViT-large encoder", 7168, 1024, 4096),
                       ("V-JEPA encoder", 20480, 768, 3072), ("ranged decoder MLP", 13312, 384, 1536)]:
    mk = lambda n: torch.randn(M, n, generator=g, device="cuda").bfloat16()
    pairs = [(mk(w), mk(ff)), (mk(ff), mk(w)), (mk(w), mk(w)), (mk(3 * w), mk(w))]
    outs = [torch.empty(X.shape[1], Y.shape[1], dtype=torch.float32, device="cuda") for X, Y in pairs]
    flop = sum(2.0 * M * X.shape[1] * Y.shape[1] for X, Y in pairs)
    raw.smbv_debug_set_variant(0)
    t_single = timeit(lambda: [ops.gemm_tn_bias(X, Y) for X, Y in pairs])
    raw.smbv_debug_set_variant(variant)
    t_group = timeit(lambda: ops.gemm_tn_grouped(pairs, sums=[1, 1, 1, 3], out=outs))
    raw.smbv_debug_set_variant(0)
    tiles256 = all(X.shape[1] % 256 == 0 and Y.shape[1] % 256 == 0 for X, Y in pairs) and not (variant & 32768)
    t = 256 if tiles256 else 128
    nblk = sum((X.shape[1] // t) * (Y.shape[1] // t) for X, Y in pairs)
    print(f"{name:20s} M {M:6d} w {w:5d}: four launches {t_single:7.1f} us ({flop / t_single / 1e6:6.0f} TF/s)   grouped {t_group:7.1f} us ({flop / t_group / 1e6:6.0f} TF/s) "
          f"{nblk} blocks of {t}x{t}: {flop / nblk / t_group / 1e6:5.2f} TF/s per block")
