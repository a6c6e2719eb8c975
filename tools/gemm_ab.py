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
variants = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 131072, 0]   # first generation, ring, shipped selection
shapes = [(7168, 2304, 768, {}), (7168, 768, 2304, {}), (7168, 768, 768, {}), (7168, 3072, 768, dict(gelu=True, save_pre=True)), (7168, 768, 3072, {}),
          (7168, 3072, 768, dict(dgelu=True)),
          (20480, 1152, 384, {}), (20480, 384, 1152, {}), (20480, 384, 384, {}), (20480, 384, 1536, {}), (20480, 1536, 384, dict(gelu=True, save_pre=True)),
          (20480, 1536, 384, dict(dgelu=True)), (13312, 4096, 384, {}), (13312, 384, 4096, {}), (7168, 768, 4096, {}), (7168, 384, 768, {})]
if len(sys.argv) > 2 and sys.argv[2] == "large":   # BASELINE config 5's model: encoder 1024 / 4096, decoder 512 / 2048
    shapes = [(7168, 3072, 1024, {}), (7168, 1024, 1024, {}), (7168, 4096, 1024, dict(gelu=True, save_pre=True)), (7168, 1024, 4096, {}), (7168, 4096, 1024, dict(dgelu=True)),
              (7168, 1024, 3072, {}), (20480, 1536, 512, {}), (20480, 512, 512, {}), (20480, 2048, 512, dict(gelu=True, save_pre=True)), (20480, 512, 2048, {}),
              (20480, 2048, 512, dict(dgelu=True)), (20480, 512, 1536, {}), (13312, 4096, 512, {}), (13312, 512, 4096, {}), (7168, 512, 1024, {})]
if len(sys.argv) > 2 and sys.argv[2] == "vjepa":   # BASELINE config 4: ViT-base encoder over all 20,480 tokens, predictor 384 / 1536 over 20,480 rows
    shapes = [(20480, 2304, 768, {}), (20480, 768, 768, {}), (20480, 3072, 768, dict(gelu=True, save_pre=True)), (20480, 768, 3072, {}), (20480, 3072, 768, dict(dgelu=True)),
              (20480, 768, 2304, {}), (20480, 1152, 384, {}), (20480, 384, 384, {}), (20480, 1536, 384, dict(gelu=True, save_pre=True)), (20480, 384, 1536, {}),
              (20480, 1536, 384, dict(dgelu=True)), (20480, 384, 1152, {}), (20480, 384, 768, {}), (12222, 768, 384, {})]
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K, kw in shapes:
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    if kw.pop("dgelu", False):
        kw = dict(dgelu_pre=torch.randn(M, N, generator=g, device="cuda").bfloat16())
    outs = {}
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            raw.smbv_debug_set_variant(v)
            r = ops.gemm_nt(A, B, bias=bias, **kw); torch.cuda.synchronize()
            outs[v] = r["bf16"]
            res[v].append(timeit(lambda: ops.gemm_nt(A, B, bias=bias, reuse=r, **kw)))
    same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants)
    f = 2.0 * M * N * K
    print(f"M{M} N{N} K{K} {'gelu' if 'gelu' in kw else ('dgel' if kw else '    ')}: " + "  ".join(f"v{v}: {sorted(res[v])[2]*1e3:7.1f}us {f/sorted(res[v])[2]/1e9:6.1f}TF" for v in variants) + f"  same={same}")
raw.smbv_debug_set_variant(0)
