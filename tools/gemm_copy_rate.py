"""Target of tools/gemm_copy_rate.sh (rocprofv3 --pmc passes): the encoder's N = 768 NT products as the step launches them (gemm_nt_dma_kernel, mixed 128 x 128 /
128 x 64 tiles, 504 blocks) - out-projection 7168 x 768 x 768, d_qkv -> d_ln 7168 x 768 x 2304, fc2 / d_fc1 7168 x 768 x 3072 - and the ring kernel's best case
(QKV 7168 x 2304 x 768), 6 launches each with preallocated outputs."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import _lib
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in ((7168, 768, 768), (7168, 768, 2304), (7168, 768, 3072), (7168, 2304, 768)):
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    e = _lib.GemmEpi()
    e.alpha = 1.0
    e.bias = bias.data_ptr()
    e.out_bf16, e.ld_ob = out.data_ptr(), N
    for _ in range(6):
        assert lib.smbv_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, M, N, K, C.byref(e), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
