"""Ablation timings of gemm_nt_ring_kernel (debug bits in smbv_debug_set_variant >> 20: 1 no epilogue, 2 no in-loop copies, 4 no MFMAs, 8 no LDS reads).
Outputs are preallocated: the timed region holds the GEMM launches only."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)

def make(M, N, K, gelu):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    pre = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    e = _lib.GemmEpi()
    e.alpha = 1.0
    e.bias = bias.data_ptr()
    e.out_bf16, e.ld_ob = out.data_ptr(), N
    if gelu:
        e.act_gelu, e.aux_pre, e.ld_aux = 1, pre.data_ptr(), N
    st = torch.cuda.current_stream().cuda_stream
    def run():
        rc = lib.smbv_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, C.byref(e), st)
        assert rc == 0
    return run, (A, B, bias, out, pre, e)

def timeit(fn, iters=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

shapes = [(7168, 2304, 768, False), (20480, 1536, 384, True), (7168, 768, 3072, False), (7168, 768, 768, False)]
for M, N, K, gelu in shapes:
    run, keep = make(M, N, K, gelu)
    raw.smbv_debug_set_variant(256)
    t_old = timeit(run)
    print(f"M{M} N{N} K{K} first-gen 128x128: {t_old:6.1f} us  {2.0*M*N*K/t_old/1e6:6.1f} TF", flush=True)
    for base, name in ((131072, "NS3"), (131072 | 512, "NS4")):
        row = []
        for dbg in (0, 1, 2, 4, 8, 2 | 1, 4 | 1, 2 | 4 | 8 | 1, 4 | 8 | 1, 2 | 8 | 1, 2 | 4 | 1):
            raw.smbv_debug_set_variant(base | (dbg << 20))
            row.append(f"d{dbg}:{timeit(run):6.1f}")
        raw.smbv_debug_set_variant(0)
        print(f"M{M} N{N} K{K} {name}: " + " ".join(row), flush=True)
