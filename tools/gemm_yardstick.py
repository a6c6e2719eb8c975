"""Yardstick only (never on the product path): the vendor library's bf16 NT GEMM (torch.mm -> hipBLASLt / rocBLAS) at the step's shapes,
beside smbv_gemm_nt without epilogue extras.  Says how far the hand-written kernels are from a tuned library at these small-K shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
def timeit(fn, iters=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(7168, 768, 768), (7168, 768, 2304), (7168, 768, 3072), (7168, 2304, 768), (7168, 3072, 768),
          (20480, 384, 384), (20480, 384, 1152), (20480, 384, 1536), (20480, 1152, 384), (20480, 1536, 384), (13312, 4096, 384)]
for M, N, K in shapes:
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    W = torch.randn(N, K, generator=g, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lib = timeit(lambda: torch.mm(A, W.t(), out=out))
    r = ops.gemm_nt(A, W, out_bf16=True)
    t_own = timeit(lambda: ops.gemm_nt(A, W, out_bf16=True, reuse=r))
    fl = 2.0 * M * N * K
    print(f"M{M:6d} N{N:5d} K{K:5d}: library {t_lib*1e3:7.1f} us {fl/t_lib/1e9:6.0f} TF | smbv_gemm_nt {t_own*1e3:7.1f} us {fl/t_own/1e9:6.0f} TF", flush=True)
print("weight-gradient products  dW[N1, N2] = X[M, N1]^T Y[M, N2]  (library: bf16 output; smbv_gemm_tn: fp32 output + slab reduction)")
for M, N1, N2 in [(7168, 2304, 768), (7168, 768, 768), (7168, 3072, 768), (7168, 768, 3072), (20480, 1152, 384), (20480, 384, 384), (20480, 1536, 384), (20480, 384, 1536),
                  (13312, 4096, 384)]:
    X = torch.randn(M, N1, generator=g, device="cuda").bfloat16()
    Y = torch.randn(M, N2, generator=g, device="cuda").bfloat16()
    out = torch.empty(N1, N2, device="cuda", dtype=torch.bfloat16)
    t_lib = timeit(lambda: torch.mm(X.t(), Y, out=out))
    t_own = timeit(lambda: ops.gemm_tn(X, Y))
    fl = 2.0 * M * N1 * N2
    print(f"M{M:6d} N1 {N1:5d} N2 {N2:5d}: library {t_lib*1e3:7.1f} us {fl/t_lib/1e9:6.0f} TF | smbv_gemm_tn {t_own*1e3:7.1f} us {fl/t_own/1e9:6.0f} TF", flush=True)
