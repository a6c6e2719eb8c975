"""Randomised shape sweep of the entry points round 4 added: the attention forward's residual output and the backward that consumes it, the fp8
recompute products of the backward (head_dim 64), and the grouped weight-gradient launch.  References: fp32 torch.  The device status word must stay 0.
usage: fuzz_round4.py [n_cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def rel(a, b): return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))
def rms(a, b): return float((a.float() - b.float()).pow(2).mean().sqrt() / (b.float().pow(2).mean().sqrt() + 1e-6))   # (+1e-6: S = 1 has exactly zero dQ / dK)
bad = 0
for i in range(n_cases):
    B, H = rng.choice([1, 1, 2]), rng.choice([1, 2, 3, 6])
    S = rng.choice([rng.randint(1, 200), rng.randint(200, 1500), 64 * rng.randint(1, 30), 128 * rng.randint(1, 20) + rng.choice([0, 1, 31, 33, 63, 65, 127])])
    if rng.random() < 0.4 and S > 8:
        Sq = rng.randint(1, S); qoff = rng.randint(0, S - Sq)
    else:
        Sq, qoff = 0, 0
    fp8 = rng.random() < 0.5
    Dh = 64 if fp8 else rng.choice([64, 32])
    w = H * Dh
    g = torch.Generator(device="cuda").manual_seed(i)
    amp = rng.choice([0.5, 1.0, 2.0])
    qkv = (torch.randn(B * S, 3 * w, generator=g, device="cuda") * amp).bfloat16()
    O, lse, O_lo = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, resid=True)
    qf = qkv.float().requires_grad_(True)
    q, k, v = (qf[:, j * w:(j + 1) * w].view(B, S, H, Dh).transpose(1, 2) for j in range(3))
    s = q @ k.transpose(-1, -2) * Dh ** -0.5
    sq, qo = (Sq, qoff) if Sq else (S, 0)
    s = s[:, :, qo:qo + sq]
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, sq, w)
    hi = O.view(B, S, w)[:, qo:qo + sq].float()
    both = hi + O_lo.view(B, S, w)[:, qo:qo + sq].float()
    dO = torch.zeros(B, S, w, device="cuda"); dO[:, qo:qo + sq] = torch.randn(B, sq, w, generator=g, device="cuda")
    dO = dO.bfloat16().view(B * S, w)
    dqkv, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, O_lo=O_lo, fp8=fp8)
    dqkv2, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, O_lo=O_lo, fp8=fp8)
    ref.backward(dO.float().view(B, S, w)[:, qo:qo + sq])
    e = [rms(dqkv[:, j * w:(j + 1) * w], qf.grad[:, j * w:(j + 1) * w]) for j in range(3)]
    # residual: hi + lo no further from the fp32 reference than hi alone (P is still rounded to bf16, so it is not 2^-8 closer); backward: rms bounds of
    # tests/test_ops_gpu.py (bf16) and tests/test_fp8_gpu.py (fp8: dQ through the e4m3 S / dP - 1.2e-1 up to unit variance, 2.5e-1 for peaky scores -
    # dK / dV untouched)
    bq = (2.5e-1 if amp > 1 else 1.2e-1) if fp8 else 1.5e-2
    few = 1.5 if sq < 64 else 1.0   # (an rms over one to a few dozen query rows scatters: peaky cases sit at 1.0-1.3e-2 / 2.0-2.2e-1 with thousands of rows and reach 1.8e-2 / 2.5e-1 with one)
    ok = rms(both, ref) <= rms(hi, ref) * 1.02 + 1e-6 and e[0] < bq * few and e[1] < 1.5e-2 * few and e[2] < 1.5e-2 * few and bool(torch.isfinite(dqkv.float()).all()) \
        and torch.equal(dqkv, dqkv2)
    bad += not ok
    print(f"attn{' fp8' if fp8 else ''} amp {amp} Dh{Dh} B{B} S{S} H{H} Sq{Sq} qoff{qoff}: O hi {rms(hi, ref):.1e} hi+lo {rms(both, ref):.1e} dq {e[0]:.1e} dk {e[1]:.1e} dv {e[2]:.1e} "
          f"{'ok' if ok else 'FAIL'}", flush=True)
for i in range(n_cases):
    n = rng.randint(1, 6)
    M = rng.choice([rng.randint(1, 300), rng.randint(300, 9000), 128 * rng.randint(1, 60)])
    wide = rng.random() < 0.5   # all widths multiples of 256 -> the 256 x 256 tile build; else multiples of 128
    unit = 256 if wide else 128
    g = torch.Generator(device="cuda").manual_seed(2000 + i)
    pairs, sums = [], []
    for j in range(n):
        N1, N2 = unit * rng.randint(1, 6 if wide else 12), unit * rng.randint(1, 6 if wide else 12)
        # operands as column slices of wider matrices (as the step passes them: q / k / v gradients of one matrix)
        padx, pady = 8 * rng.randint(0, 4), 8 * rng.randint(0, 4)
        Xw = torch.randn(M, N1 + padx, generator=g, device="cuda").bfloat16(); Yw = torch.randn(M, N2 + pady, generator=g, device="cuda").bfloat16()
        pairs.append((Xw[:, padx:], Yw[:, :N2]))
        nseg = rng.choice([0, 0, 1, 3]) if N1 % (3 * 8) == 0 else rng.choice([0, 1])
        sums.append(nseg)
    outs, xs = ops.gemm_tn_grouped(pairs, sums=sums)
    es, ok = [], True
    for (X, Y), o, seg, nseg in zip(pairs, outs, xs, sums):
        e2 = rel(o, X.float().t() @ Y.float())
        es.append(e2)
        ok = ok and e2 < 1e-4
        if nseg:
            cs = X.float().sum(0)
            got = torch.cat(seg)
            ok = ok and rel(got, cs) < 1e-4
    bad += not ok
    print(f"grouped M{M} jobs {[(p[0].shape[1], p[1].shape[1], s) for p, s in zip(pairs, sums)]}: worst {max(es):.1e} {'ok' if ok else 'FAIL'}", flush=True)
st = ops.device_status()
print("device status", st)
print("FUZZ", "CLEAN" if bad == 0 and st == 0 else f"FAILED ({bad})")
