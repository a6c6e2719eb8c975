"""Target of rocprofv3 --pmc passes on the NT GEMM kernels: the encoder QKV product (7168 x 2304 x 768) and the decoder fc1 + GELU product
(20480 x 1536 x 384), first-generation 128 x 128 kernel and ring kernel, 5 launches each (preallocated outputs)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K, gelu in ((7168, 2304, 768, False), (20480, 1536, 384, True)):
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
    for variant in (256, 131072):
        lib.smbv_debug_set_variant(variant)
        for _ in range(5):
            assert lib.smbv_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, C.byref(e), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
    lib.smbv_debug_set_variant(0)
