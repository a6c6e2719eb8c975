"""Randomised shape sweep of the attention and GEMM entry points against fp32 torch references (tolerances of tests/test_ops_gpu.py).
usage: fuzz_ops.py [n_cases] [seed] [debug_variant]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
if len(sys.argv) > 3:
    from smb_vision_amd import _lib
    _lib.load().smbv_debug_set_variant(int(sys.argv[3]))
def rel(a, b): return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
bad = 0
for i in range(n_cases):
    B, H = rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 3, 5])
    S = rng.choice([rng.randint(1, 200), rng.randint(200, 1500), 64 * rng.randint(1, 30), 128 * rng.randint(1, 20) + rng.choice([0, 1, 31, 33, 63, 65, 127])])
    if rng.random() < 0.4 and S > 8:
        Sq = rng.randint(1, S); qoff = rng.randint(0, S - Sq)
    else:
        Sq, qoff = 0, 0
    Dh = rng.choice([64, 32])
    w = H * Dh
    g = torch.Generator(device="cuda").manual_seed(i)
    qkv = (torch.randn(B * S, 3 * w, generator=g, device="cuda") * rng.choice([0.5, 1.0, 2.0])).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    qf = qkv.float().requires_grad_(True)
    q, k, v = (qf[:, j * w:(j + 1) * w].view(B, S, H, Dh).transpose(1, 2) for j in range(3))
    s = q @ k.transpose(-1, -2) * Dh ** -0.5
    sq, qo = (Sq, qoff) if Sq else (S, 0)
    s = s[:, :, qo:qo + sq]
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, sq, w)
    got = O.view(B, S, w)[:, qo:qo + sq]
    dO = torch.zeros(B, S, w, device="cuda"); dO[:, qo:qo + sq] = torch.randn(B, sq, w, generator=g, device="cuda")
    dO = dO.bfloat16().view(B * S, w)
    dqkv, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    ref.backward(dO.float().view(B, S, w)[:, qo:qo + sq])
    e = [rel(got, ref), rel(lse, torch.logsumexp(s, -1))] + [rel(dqkv[:, j * w:(j + 1) * w], qf.grad[:, j * w:(j + 1) * w]) for j in range(3)]
    rms = max(float((dqkv[:, j * w:(j + 1) * w].float() - qf.grad[:, j * w:(j + 1) * w]).pow(2).mean().sqrt() / qf.grad[:, j * w:(j + 1) * w].pow(2).mean().sqrt()) for j in range(3))
    # max-abs bound 5e-2 (single elements of a short query range carry the bf16 rounding of dS un-averaged), rms bound 1.5e-2 as in tests/test_ops_gpu.py
    ok = e[0] < 2.5e-2 and e[1] < 4e-3 and max(e[2:]) < 5e-2 and rms < 1.5e-2 and bool(torch.isfinite(dqkv.float()).all())
    bad += not ok
    print(f"attn Dh{Dh} B{B} S{S} H{H} Sq{Sq} qoff{qoff}: O {e[0]:.1e} lse {e[1]:.1e} dq {e[2]:.1e} dk {e[3]:.1e} dv {e[4]:.1e} rms {rms:.1e} {'ok' if ok else 'FAIL'}", flush=True)
for i in range(n_cases):
    M = rng.choice([rng.randint(1, 300), rng.randint(300, 9000)])
    N, K = 8 * rng.randint(1, 400), 8 * rng.randint(1, 260)
    g = torch.Generator(device="cuda").manual_seed(1000 + i)
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    Bm = (torch.randn(N, K, generator=g, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, generator=g, device="cuda")
    r = ops.gemm_nt(A, Bm, bias=bias)
    ref = A.float() @ Bm.float().t() + bias
    e1 = rel(r["bf16"], ref)
    N1, N2 = 8 * rng.randint(1, 300), 8 * rng.randint(1, 300)
    X = torch.randn(M, N1, generator=g, device="cuda").bfloat16(); Y = torch.randn(M, N2, generator=g, device="cuda").bfloat16()
    e2 = rel(ops.gemm_tn(X, Y), X.float().t() @ Y.float())
    ok = e1 < 1e-2 and e2 < 1e-4
    bad += not ok
    print(f"gemm M{M} N{N} K{K} nt {e1:.1e} | tn N1 {N1} N2 {N2} {e2:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"FAILED ({bad})")
