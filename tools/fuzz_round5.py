"""Randomised shape sweep of the entry points round 5 added (ABI 3003 / 3004), each against the path it replaces:
  * smbv_gemm_nt_rope             == smbv_gemm_nt followed by smbv_rope3d, BITWISE (ring-kernel shapes and shapes that fall back to the two launches)
  * smbv_attention_bwd_rope       == smbv_attention_bwd(_split) followed by smbv_rope3d(transpose), BITWISE (head slots 64 / 32, real head_dim below the
                                     slot, ragged S, query sub-ranges, position lists and row % S, with and without half sweeps)
  * smbv_gemm_tn_grouped_split    == smbv_gemm_tn_grouped up to the order of two fp32 additions (2e-6), identical run to run
The device status word must stay 0.   usage: fuzz_round5.py [n_cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smb_vision_amd import ops, _lib
from smb_vision_amd.modeling_vjepa import rope_table

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
bad = 0


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def rope_case(rows, S, Dh, hs, g):
    grid = rng.choice([2, 4, 8, 14])
    frames = rng.randint(1, 20)
    P = max(frames, grid)
    with_pos = rng.random() < 0.6 or S * 1 > frames * grid * grid
    pos = torch.randint(0, frames * grid * grid, (rows,), generator=g, device="cuda").to(torch.int32) if with_pos else None
    rows_known = rng.random() < 0.85
    return dict(table=rope_table(Dh, P + 1, "cuda"), pos=pos, S=S, grid=grid, Dh=Dh, head_stride=hs, **({} if rows_known else dict(table_rows=0)))


for i in range(n_cases):
    g = torch.Generator(device="cuda").manual_seed(1000 + i)
    # ---- QKV product with the rotary epilogue
    H = rng.choice([1, 2, 3, 6, 12])
    hs = rng.choice([32, 64])
    Dh = rng.choice([d for d in (8, 16, 24, 32, 40, 48, 64) if d <= hs])
    wa = H * hs
    K = rng.choice([128, 256, 384, 768])
    S = rng.randint(16, 1500)
    Bq = rng.choice([1, 1, 2])
    M = Bq * S
    A = torch.randn(M, K, generator=g, device="cuda").bfloat16()
    Wt = (torch.randn(3 * wa, K, generator=g, device="cuda") / K ** 0.5).bfloat16()
    bias = 0.1 * torch.randn(3 * wa, generator=g, device="cuda")
    rope = rope_case(M, S, Dh, hs, g)
    fused = ops.gemm_nt(A, Wt, bias=bias, rope=dict(rope, wa=wa))["bf16"]
    two = ops.gemm_nt(A, Wt, bias=bias)["bf16"]
    ops.rope3d_(two, wa, H, rope)
    ok1 = torch.equal(fused, two)
    # ---- attention backward storing through the transposed map
    slot = rng.choice([64, 64, 32])
    real = rng.choice([d for d in (8, 16, 24, 32, 40, 56, 64) if d <= slot])
    Ha = rng.choice([1, 2, 3, 6])
    Ba = rng.choice([1, 1, 2])
    Sa = rng.choice([rng.randint(1, 300), rng.randint(300, 2500), 128 * rng.randint(1, 24) + rng.choice([0, 1, 63, 65, 127])])
    if rng.random() < 0.35 and Sa > 8:
        Sq = rng.randint(1, Sa); qoff = rng.randint(0, Sa - Sq)
    else:
        Sq, qoff = 0, 0
    w = Ha * slot
    qkv = torch.randn(Ba * Sa, 3 * w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, Ba, Sa, Ha, Sq=Sq, q_off=qoff, Dh=slot)
    n = Sq if Sq > 0 else Sa
    dO = torch.zeros(Ba, Sa, w, device="cuda")
    dO[:, qoff:qoff + n] = torch.randn(Ba, n, w, generator=g, device="cuda")
    dO = dO.bfloat16().view(Ba * Sa, w)
    rope2 = rope_case(Ba * Sa, Sa, real, slot, g)
    need = lib.smbv_attention_bwd_split_workspace_bytes(Ba, Sa, Ha, slot)
    split = torch.empty(max(need, 256), dtype=torch.uint8, device="cuda") if (need > 0 and rng.random() < 0.7) else None
    two_b, _ = ops.attention_bwd(qkv, O, dO, lse, Ba, Sa, Ha, Sq=Sq, q_off=qoff, Dh=slot, split=split)
    ops.rope3d_(two_b, w, Ha, rope2, transpose=True)
    fused_b, _ = ops.attention_bwd(qkv, O, dO, lse, Ba, Sa, Ha, Sq=Sq, q_off=qoff, Dh=slot, split=split, rope=rope2)
    ok2 = torch.equal(fused_b, two_b)
    # ---- grouped weight gradients, token halves
    nj = rng.randint(1, 4)
    shapes = []
    tiles = 0
    for _ in range(nj):
        n1, n2 = 256 * rng.randint(1, 4), 256 * rng.randint(1, 4)
        if tiles + (n1 // 256) * (n2 // 256) > 128:
            break
        tiles += (n1 // 256) * (n2 // 256)
        shapes.append((rng.randint(256, 3000), n1, n2))
    pairs = [(torch.randn(Mj, n1, generator=g, device="cuda").bfloat16(), torch.randn(Mj, n2, generator=g, device="cuda").bfloat16()) for Mj, n1, n2 in shapes]
    nseg = [rng.choice([1, 1, 3]) if (n1 % 3 == 0 and (n1 // 3) % 256 == 0) else 1 for _, n1, _ in shapes]
    whole, xw = ops.gemm_tn_grouped(pairs, sums=nseg)
    sp1, xs1 = ops.gemm_tn_grouped(pairs, sums=nseg, split=True)
    sp2, xs2 = ops.gemm_tn_grouped(pairs, sums=nseg, split=True)
    e3 = max(max(rel(a, b) for a, b in zip(sp1, whole)), max(rel(torch.cat(a), torch.cat(b)) for a, b in zip(xs1, xw)))
    ok3 = e3 < 2e-6 and all(torch.equal(a, b) for a, b in zip(sp1, sp2)) and all(torch.equal(torch.cat(a), torch.cat(b)) for a, b in zip(xs1, xs2))
    torch.cuda.synchronize()
    st = ops.device_status()
    good = ok1 and ok2 and ok3 and st == 0
    bad += not good
    print(f"case {i:3d} {'ok ' if good else 'BAD'} | qkv M{M} K{K} H{H} hs{hs} Dh{Dh} grid{rope['grid']} pos{int(rope['pos'] is not None)} rows{int('table_rows' not in rope)}: {ok1}"
          f" | attn B{Ba} S{Sa} H{Ha} slot{slot} real{real} Sq{Sq}+{qoff} split{int(split is not None)} pos{int(rope2['pos'] is not None)}: {ok2}"
          f" | tn {shapes} sums{nseg}: {e3:.1e} {ok3} | status {st}", flush=True)
print(f"{n_cases - bad} / {n_cases} cases agree" + ("" if bad == 0 else "  <-- MISMATCHES"))
sys.exit(1 if bad else 0)
