import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
_lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 32]
shapes = [(7168, 3072, 768), (7168, 768, 3072), (7168, 2304, 768), (7168, 768, 768), (20480, 1536, 384), (20480, 384, 1536), (20480, 1152, 384),
          (20480, 384, 384), (13312, 4096, 384), (7168, 768, 4096), (7168, 384, 768), (7000, 1024, 256), (300, 256, 512)]
g = torch.Generator(device="cuda").manual_seed(0)
for M, N1, N2 in shapes:
    X = torch.randn(M, N1, generator=g, device="cuda").bfloat16()
    Y = torch.randn(M, N2, generator=g, device="cuda").bfloat16()
    outs, res = {}, {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            raw.smbv_debug_set_variant(v)
            outs[v] = ops.gemm_tn(X, Y); torch.cuda.synchronize()
            res[v].append(timeit(lambda: ops.gemm_tn(X, Y)))
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
    ref = X.float().T @ Y.float()
    err = "  ".join(f"v{v}:{float((outs[v] - ref).abs().max() / ref.abs().max()):.1e}" for v in variants)
    f = 2.0 * M * N1 * N2
    print(f"M{M} N1 {N1} N2 {N2}: " + "  ".join(f"v{v}: {sorted(res[v])[2]*1e3:7.1f}us {f/sorted(res[v])[2]/1e9:6.1f}TF" for v in variants) + f"  same={same}  max rel err vs fp32 {err}")
raw.smbv_debug_set_variant(0)
