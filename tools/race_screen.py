"""Race screen for the hand-scheduled kernels at the bench shapes: N repeated launches must agree bit for bit with the first
(attention fwd / bwd at the encoder and decoder shapes, TN and NT GEMMs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for name, S, H, Sq, qoff, Dh in (("enc", 7168, 12, 0, 0, 64), ("dec", 20480, 6, 0, 0, 64), ("dec-tail", 20480, 6, 13312, 7168, 64), ("ragged", 5003, 3, 0, 0, 64),
                                 ("predictor", 20480, 12, 0, 0, 32), ("ragged-32", 5003, 3, 0, 0, 32)):
    w = H * Dh
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(S, 3 * w, generator=g, device="cuda").bfloat16()
    dO = torch.randn(S, w, generator=g, device="cuda").bfloat16()
    O0, l0 = ops.attention_fwd(qkv, 1, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    d0, _ = ops.attention_bwd(qkv, O0, dO, l0, 1, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    for i in range(N):
        O, l = ops.attention_fwd(qkv, 1, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        d, _ = ops.attention_bwd(qkv, O0, dO, l0, 1, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        if not (torch.equal(O, O0) and torch.equal(l, l0) and torch.equal(d, d0)):
            bad += 1
            print(f"{name}: run {i} differs (fwd {torch.equal(O, O0)}, lse {torch.equal(l, l0)}, bwd {torch.equal(d, d0)})", flush=True)
    print(f"{name}: {N} runs checked", flush=True)
for M, N1, N2 in ((7168, 3072, 768), (20480, 384, 1536)):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(M, N1, generator=g, device="cuda").bfloat16()
    Y = torch.randn(M, N2, generator=g, device="cuda").bfloat16()
    r0 = ops.gemm_tn(X, Y).clone()
    for i in range(N):
        if not torch.equal(ops.gemm_tn(X, Y), r0):
            bad += 1
            print(f"gemm_tn {M}x{N1}x{N2}: run {i} differs", flush=True)
    print(f"gemm_tn {M}x{N1}x{N2}: {N} runs checked", flush=True)
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
