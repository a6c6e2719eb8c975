"""GPU parity tests of the individual HIP kernels, called through the C ABI (smb_vision_amd.ops -> libsmbv.so).

Each kernel is compared with a plain PyTorch fp32 computation of the same op on the same (bf16-rounded) operands.
Tolerances: the kernels take bf16 operands, accumulate in fp32 and (mostly) round the result to bf16, so an output
element may be off by one bf16 ulp (2^-8 relative) of its magnitude plus fp32 summation-order noise.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from smb_vision_amd import ops as o
    return o


def rel_err(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def rms_err(a, b):
    return float((a.float() - b.float()).pow(2).mean().sqrt() / (b.float().pow(2).mean().sqrt() + 1e-30))


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1, 8, 8), (1000, 768, 768), (333, 2304, 768), (26, 4096, 64)])
def test_gemm_nt_plain(ops, M, N, K):
    A = rnd(M, K, seed=1).bfloat16()
    B = rnd(N, K, seed=2).bfloat16()
    bias = rnd(N, seed=3)
    r = ops.gemm_nt(A, B, bias=bias, out_bf16=True, out_f32=True)
    ref = A.float() @ B.float().t() + bias
    assert rel_err(r["f32"], ref) < 1e-5 * math.sqrt(K) + 1e-6
    assert rel_err(r["bf16"], ref) < 6e-3


def test_gemm_nt_asymmetric_identity(ops):
    # A = I with an asymmetric B catches a transposed store
    A = torch.eye(128, device="cuda").bfloat16()
    B = (torch.arange(128 * 128, device="cuda").reshape(128, 128) % 251).float().bfloat16()
    r = ops.gemm_nt(A, B, out_bf16=False, out_f32=True)
    assert torch.equal(r["f32"], B.float().t().contiguous())


def test_gemm_nt_epilogues(ops):
    M, N, K = 300, 256, 128
    A = rnd(M, K, seed=1).bfloat16()
    B = rnd(N, K, seed=2, scale=0.1).bfloat16()
    bias = rnd(N, seed=3)
    resid = rnd(M, N, seed=4)
    # bias + residual -> fp32
    r = ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True)
    ref = A.float() @ B.float().t() + bias + resid
    assert rel_err(r["f32"], ref) < 1e-4
    # gathered row add (position table) + output row remap
    table = rnd(50, N, seed=5)
    idx = torch.randint(0, 50, (M,), device="cuda", dtype=torch.int32)
    r = ops.gemm_nt(A, B, rowadd=table, rowidx=idx, out_bf16=False, out_f32=True, grp=(100, 150), out_rows=450)
    ref = A.float() @ B.float().t() + table[idx.long()]
    out = r["f32"].view(3, 150, N)[:, :100].reshape(M, N)
    assert rel_err(out, ref) < 1e-4
    # GELU with saved pre-activation
    r = ops.gemm_nt(A, B, bias=bias, gelu=True, save_pre=True)
    pre = (A.float() @ B.float().t() + bias)
    assert rel_err(r["pre"], pre) < 6e-3
    ref = torch.nn.functional.gelu(r["pre"].float())
    assert rel_err(r["bf16"], ref) < 6e-3
    # dgrad through GELU
    pre_b = rnd(M, N, seed=7).bfloat16()
    r = ops.gemm_nt(A, B, dgelu_pre=pre_b)
    x = pre_b.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = (A.float() @ B.float().t()) * x.grad
    assert rel_err(r["bf16"], ref) < 8e-3


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (300, 136, 72), (26, 64, 4096), (4096, 128, 128), (1000, 768, 256), (7, 8, 8)])
def test_gemm_tn(ops, M, N1, N2):
    X = rnd(M, N1, seed=1).bfloat16()
    Y = rnd(M, N2, seed=2).bfloat16()
    out = ops.gemm_tn(X, Y)
    ref = X.float().t() @ Y.float()
    assert rel_err(out, ref) < 1e-5 * math.sqrt(M) + 1e-6


@pytest.mark.parametrize("M,N1,N2", [(2100, 2048, 1024), (7168, 768, 3072), (4100, 3072, 768)])
def test_gemm_tn_wide_outputs_take_the_256_tile_kernel(ops, M, N1, N2):
    """>= 32 tiles of 256 x 256 with slabs of >= 256 tokens: gemm_tn_256_kernel (ping-pong wave groups, 4-slot hidden LDS-DMA ring); ragged last slab;
    an identity block in X pins the operand orientation (out[n1, n2] = sum_m X[m, n1] Y[m, n2])."""
    X = rnd(M, N1, seed=5).bfloat16()
    Y = rnd(M, N2, seed=6).bfloat16()
    out = ops.gemm_tn(X, Y)
    ref = X.float().t() @ Y.float()
    assert rel_err(out, ref) < 1e-5 * math.sqrt(M) + 1e-6
    assert torch.equal(out, ops.gemm_tn(X, Y))  # run-to-run identical (slabs summed in slab order)
    Xi = torch.zeros(M, N1, device="cuda").bfloat16()
    Xi[:N1 if N1 < M else M].fill_diagonal_(1.0)
    n = min(N1, M)
    assert torch.equal(ops.gemm_tn(Xi, Y)[:n], Y.float()[:n])


def test_gemm_tn_asymmetric(ops):
    M = 128
    X = torch.eye(M, device="cuda").bfloat16()                       # X^T Y = Y
    Y = (torch.arange(M * 64, device="cuda").reshape(M, 64) % 127).float().bfloat16()
    assert torch.equal(ops.gemm_tn(X, Y), Y.float())


@pytest.mark.parametrize("shapes", [
    [(1000, 256, 512), (1000, 512, 256), (777, 256, 256), (1000, 768, 256)],     # every width a multiple of 256: 256 x 256 tiles, ragged token counts
    [(2100, 384, 1536), (2100, 1536, 384), (1300, 384, 384), (2100, 1152, 384)],  # decoder widths: 128 x 128 tiles
    [(33, 128, 128)],
])
def test_gemm_tn_grouped_is_the_products_one_by_one(ops, shapes):
    """smbv_gemm_tn_grouped: a block's weight-gradient products as one launch, each tile reduced over all tokens (no slabs), bias sums written directly
    (three segments for the fused q | k | v product, one of them dropped as VideoMAE's missing key bias is)."""
    pairs = [(rnd(M, N1, seed=10 + i).bfloat16(), rnd(M, N2, seed=20 + i).bfloat16()) for i, (M, N1, N2) in enumerate(shapes)]
    nseg = [3 if N1 % 3 == 0 and (N1 // 3) % 128 == 0 else 1 for (_, N1, _) in shapes]
    outs, xs = ops.gemm_tn_grouped(pairs, sums=nseg)
    outs2, xs2 = ops.gemm_tn_grouped(pairs, sums=nseg)
    for (X, Y), o, o2, x, x2, (M, N1, N2) in zip(pairs, outs, outs2, xs, xs2, shapes):
        ref = X.float().t() @ Y.float()
        assert rel_err(o, ref) < 1e-5 * math.sqrt(M) + 1e-6
        assert torch.equal(o, o2)                                            # run-to-run identical
        got = torch.cat(x)
        assert rel_err(got, X.float().sum(0)) < 1e-5 * math.sqrt(M) + 1e-6 and torch.equal(got, torch.cat(x2))
        single, xsingle = ops.gemm_tn_bias(X, Y)                              # the per-product launch (slabs + finishing kernel): same sums, other order
        assert rel_err(o, single) < 2e-5 and rel_err(got, xsingle) < 2e-5
    with pytest.raises(ValueError):
        ops.gemm_tn_grouped([(rnd(64, 136, seed=1).bfloat16(), rnd(64, 128, seed=2).bfloat16())])


@pytest.mark.parametrize("shapes", [
    [(1000, 256, 512), (1000, 512, 256), (777, 256, 256), (1000, 768, 256)],       # ragged token counts: the halves are [0, (M / 2) & ~31) and the rest
    [(7168, 768, 3072), (7168, 3072, 768), (7168, 768, 768), (7168, 2304, 768)],   # the ViT-base encoder block's launch: 108 tiles, 216 workgroups
    [(256, 256, 256)],
])
def test_gemm_tn_grouped_split_hands_the_first_half_over_exactly(ops, shapes):
    """smbv_gemm_tn_grouped_split (ABI 3003): every tile's token range as two halves by two workgroups, the first handing its fp32 tile and column sums to the
    second through scratch.  Same products as the whole-range launch up to the order of two fp32 additions, identical run to run, flags left clear (the
    second and third call would otherwise read a stale hand-off or trip the bounded wait), device status word untouched."""
    pairs = [(rnd(M, N1, seed=10 + i).bfloat16(), rnd(M, N2, seed=20 + i).bfloat16()) for i, (M, N1, N2) in enumerate(shapes)]
    nseg = [3 if N1 % 3 == 0 and (N1 // 3) % 256 == 0 else 1 for (_, N1, _) in shapes]
    whole, xw = ops.gemm_tn_grouped(pairs, sums=nseg)
    runs = [ops.gemm_tn_grouped(pairs, sums=nseg, split=True) for _ in range(3)]
    torch.cuda.synchronize()
    assert ops.device_status() == 0
    outs, xs = runs[0]
    for i, ((X, Y), o, w, x, xw_i, (M, N1, N2)) in enumerate(zip(pairs, outs, whole, xs, xw, shapes)):
        ref = X.float().t() @ Y.float()
        assert rel_err(o, ref) < 1e-5 * math.sqrt(M) + 1e-6
        assert rel_err(o, w) < 2e-6
        assert rel_err(torch.cat(x), torch.cat(xw_i)) < 2e-6
        for other, xo in runs[1:]:
            assert torch.equal(o, other[i]) and torch.equal(torch.cat(x), torch.cat(xo[i]))
    with pytest.raises(ValueError):   # a launch the split form does not take is refused, not run whole behind the caller's back
        ops.gemm_tn_grouped([(rnd(512, 384, seed=1).bfloat16(), rnd(512, 256, seed=2).bfloat16())], split=True)
    with pytest.raises(ValueError):
        ops.gemm_tn_grouped([(rnd(100, 256, seed=1).bfloat16(), rnd(100, 256, seed=2).bfloat16())], split=True)


@pytest.mark.parametrize("rows,D", [(5, 64), (1000, 768), (333, 384), (64, 128), (17, 1024)])
def test_layernorm(ops, rows, D):
    x = rnd(rows, D, seed=1) * 2 + 0.5
    g = 1 + 0.1 * rnd(D, seed=2)
    b = 0.1 * rnd(D, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-12)
    ref = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-12)
    assert rel_err(y, ref) < 6e-3
    assert rel_err(mean, x.mean(-1)) < 1e-5
    dy = rnd(rows, D, seed=4).bfloat16()
    dres = rnd(rows, D, seed=5)
    dx, dxb, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, g, dres)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-12).backward(dy.float())
    assert rel_err(dx, xr.grad + dres) < 2e-5
    assert rel_err(dxb, xr.grad + dres) < 6e-3
    assert rel_err(dg, gr.grad) < 1e-4
    assert rel_err(db, br.grad) < 1e-4


def test_layernorm_row_remap(ops):
    # decoder.norm reads the last n_mask rows of every volume
    B, N, nmask, D = 2, 10, 6, 64
    x = rnd(B * N, D, seed=1)
    g, b = 1 + 0.1 * rnd(D, seed=2), 0.1 * rnd(D, seed=3)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5, rows=B * nmask, grp=(nmask, N, N - nmask))
    ref = torch.nn.functional.layer_norm(x.view(B, N, D)[:, N - nmask:].reshape(-1, D), (D,), g, b, 1e-5)
    assert rel_err(y, ref) < 6e-3


def _attn_ref(qkv, B, S, H, Dh=64):
    w = H * Dh
    q, k, v = (qkv.float()[:, i * w:(i + 1) * w].view(B, S, H, Dh).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) * Dh ** -0.5
    return q, k, v, s


@pytest.mark.parametrize("Dh", [64, 32])
@pytest.mark.parametrize("B,S,H", [(1, 32, 1), (2, 12, 2), (1, 100, 2), (2, 256, 1), (1, 1000, 3), (1, 2048, 2)])
def test_attention_fwd_bwd(ops, B, S, H, Dh):
    """Dh 32: the native build for the V-JEPA predictor's heads (384 / 12, modeling_vjepa.py:251-253) - 64-byte LDS rows, half the MFMA k-steps."""
    w = H * Dh
    qkv = rnd(B * S, 3 * w, seed=1).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Dh=Dh)
    qf = qkv.float().requires_grad_(True)
    q, k, v, s = _attn_ref(qf, B, S, H, Dh)
    p = torch.softmax(s, -1)
    ref = (p @ v).transpose(1, 2).reshape(B * S, w)
    assert rel_err(O, ref) < 1.5e-2
    assert rms_err(O, ref) < 6e-3
    # Q is pre-multiplied by scale*log2(e) and re-rounded to bf16 inside the kernel (one extra 2^-9 relative rounding per
    # operand, the size of the bf16 score rounding the reference's autocast matmul has anyway)
    assert rel_err(lse, torch.logsumexp(s, -1)) < 2e-3
    dO = rnd(B * S, w, seed=2).bfloat16()
    dqkv, delta = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Dh=Dh)
    ref.backward(dO.float())
    assert rel_err(delta, (dO.float() * O.float()).view(B, S, H, Dh).sum(-1).transpose(1, 2)) < 1e-4
    for i, name in enumerate("qkv"):
        a, b = dqkv[:, i * w:(i + 1) * w], qf.grad[:, i * w:(i + 1) * w]
        assert rel_err(a, b) < 2.5e-2, name
        assert rms_err(a, b) < 1e-2, name


def test_attention_online_softmax_rescale(ops):
    # spike one key per tile so the running max jumps at every tile boundary (exercises the rescale path)
    B, S, H = 1, 512, 1
    qkv = rnd(B * S, 192, seed=3).bfloat16()
    for t in range(0, S, 64):
        qkv[t + (t // 64) % 64, 64:128] *= (4 + t / 64)
    O, lse = ops.attention_fwd(qkv, B, S, H)
    q, k, v, s = _attn_ref(qkv, B, S, H)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, 64)
    # scores here reach |s| ~ 40: the in-kernel bf16 re-rounding of scale*log2e*Q perturbs them by ~|s| * 2^-9
    assert rel_err(O, ref) < 2.5e-2
    assert rel_err(lse, torch.logsumexp(s, -1)) < 4e-3


def test_attention_large_logits_stay_finite(ops):
    """Scores of +-250 (log2 domain): one missed move of the running reference overflows exp2 and poisons the row.  Every key column of
    the accumulator layout - both lane halves - has to take part in the row maximum (a compiler fold once dropped one half)."""
    B, S, H = 1, 2048, 2
    w = H * 64
    qkv = rnd(B * S, 3 * w, seed=21)
    qkv[:, :2 * w] *= 6.0
    qkv = qkv.bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H)
    assert bool(torch.isfinite(O.float()).all()) and bool(torch.isfinite(lse).all())
    q, k, v, s = _attn_ref(qkv.double(), B, S, H)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, w)
    # |s| ~ 150: the bf16 re-rounding of the pre-scaled Q / K moves scores by ~|s| * 2^-8, so compare the bulk (rms), not single rows
    assert rms_err(O, ref.float()) < 6e-2
    assert rel_err(lse, torch.logsumexp(s, -1).float()) < 8e-3
    dO = rnd(B * S, w, seed=22).bfloat16()
    dqkv, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H)
    assert bool(torch.isfinite(dqkv.float()).all())


@pytest.mark.parametrize("profile", ["rising", "falling", "spike"])
def test_attention_reference_moves_across_tiles(ops, profile):
    """Key norms that grow / shrink along the sequence or spike at one key: the row maxima move by hundreds (log2 units) between
    64-key tiles, so the forward's deferred-maximum rebase fires on many tiles (rising), only on the first (falling), or once mid-row
    (spike); a first tile far BELOW the later maxima must not leave 0 * inf behind."""
    B, S, H = 1, 640, 2
    w = H * 64
    ramp = {"rising": torch.linspace(0.05, 60.0, S), "falling": torch.linspace(60.0, 0.05, S),
            "spike": torch.where(torch.arange(S) == 400, 80.0, 0.1)}[profile]
    qkv = rnd(B * S, 3 * w, seed=31)
    qkv[:, w:2 * w] *= ramp.cuda()[:, None]
    qkv[:, :w] *= 3.0
    qkv = qkv.bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H)
    assert bool(torch.isfinite(O.float()).all()) and bool(torch.isfinite(lse).all())
    q, k, v, s = _attn_ref(qkv.double(), B, S, H)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, w)
    assert rms_err(O, ref.float()) < 6e-2  # (as above: |s| in the hundreds, bf16 re-rounding of the pre-scaled Q)
    assert rel_err(lse, torch.logsumexp(s, -1).float()) < 8e-3


@pytest.mark.parametrize("Dh", [64, 32])
def test_attention_optimistic_sweep_and_its_fallback(ops, Dh):
    """The forward first sweeps the keys WITHOUT per-tile maxima (every tile exponentiated against the first tile's row maxima) and repeats the sweep with
    the guarded tiles only if a row sum left the fp32 range (attention_fwd.hip).  (a) ordinary logits: the optimistic result agrees with the guarded-only
    kernel (debug bit 2048) to bf16 rounding; (b) logits that grow by hundreds after the first tile: the fallback must fire - finite output that is BITWISE
    the guarded-only kernel's, since the repeated sweep is that kernel's."""
    import ctypes
    from smb_vision_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    B, S, H = 2, 700, 2
    w = H * Dh

    def both(qkv):
        out = {}
        for v in (0, 2048):
            raw.smbv_debug_set_variant(v)
            try:
                out[v] = ops.attention_fwd(qkv, B, S, H, Dh=Dh)
            finally:
                raw.smbv_debug_set_variant(0)
        return out

    qkv = rnd(B * S, 3 * w, seed=41).bfloat16()
    r = both(qkv)
    q, k, v, s = _attn_ref(qkv, B, S, H, Dh)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, w)
    for key in (0, 2048):
        assert rel_err(r[key][0], ref) < 1.5e-2 and rel_err(r[key][1], torch.logsumexp(s, -1)) < 2e-3
    assert rel_err(r[0][0], r[2048][0]) < 8e-3 and float((r[0][1] - r[2048][1]).abs().max()) < 1e-4
    hot = rnd(B * S, 3 * w, seed=42)
    hot[:, w:2 * w] *= torch.linspace(0.05, 60.0, S).repeat(B).cuda()[:, None]   # key norms (and with them the logits) rise along the sequence
    hot[:, :w] *= 3.0
    r = both(hot.bfloat16())
    assert bool(torch.isfinite(r[0][0].float()).all()) and bool(torch.isfinite(r[0][1]).all())
    assert torch.equal(r[0][0], r[2048][0]) and torch.equal(r[0][1], r[2048][1])


@pytest.mark.parametrize("ps,ts", [(16, 16), (8, 8), (4, 8)])
def test_bricks_stats_mse(ops, ps, ts):
    B, T, H, W = 2, 32, 64, 64
    pix = torch.rand(B, T, 1, H, W, generator=torch.Generator().manual_seed(0)).cuda()
    gh, gw = H // ps, W // ps
    N = (T // ts) * gh * gw
    K = ts * ps * ps
    nsel = max(2, N // 3)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:nsel].sort().values for b in range(B)]).int().cuda()
    # reference patchify: conv-weight order (kt,kh,kw) == label order (ts,ps,ps,C) for C = 1
    f = pix.view(B, T // ts, ts, 1, gh, ps, gw, ps).permute(0, 1, 4, 6, 2, 5, 7, 3).reshape(B, N, K)
    sel = torch.gather(f, 1, idx.long()[..., None].expand(-1, -1, K)).reshape(B * nsel, K)
    out = ops.gather_bricks(pix, idx, nsel, ts, ps)
    assert torch.equal(out, sel.bfloat16())
    mean, rinv = ops.brick_stats(pix, idx, nsel, ts, ps)
    assert rel_err(mean, sel.mean(-1)) < 1e-5
    assert rel_err(rinv, 1 / (sel.var(-1, unbiased=True).sqrt() + 1e-6)) < 1e-4
    labels = (sel - sel.mean(-1, keepdim=True)) / (sel.var(-1, unbiased=True, keepdim=True).sqrt() + 1e-6)
    logits = rnd(B * nsel, K, seed=5).bfloat16()
    loss = ops.mse_fwd(pix, idx, nsel, mean, rinv, logits, ts, ps)
    ref = torch.nn.functional.mse_loss(logits.float(), labels)
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    gs = torch.tensor([0.5], device="cuda")
    dl = ops.mse_bwd(pix, idx, nsel, mean, rinv, logits, ts, ps, gs)
    refd = 2 * (logits.float() - labels) / labels.numel() * 0.5
    assert rel_err(dl, refd) < 6e-3


def test_casts_and_colsum(ops):
    x = rnd(96, 200, seed=1)
    assert torch.equal(ops.cast_bf16(x), x.bfloat16())
    assert torch.equal(ops.transpose_cast_bf16(x), x.t().contiguous().bfloat16())
    X = rnd(1000, 264, seed=2).bfloat16()
    assert rel_err(ops.colsum_bf16(X), X.float().sum(0)) < 1e-5


def test_errors_are_loud(ops):
    A = rnd(16, 12).bfloat16()  # K = 12 is not a multiple of 8
    with pytest.raises(ValueError):
        ops.gemm_nt(A, A)
    with pytest.raises(Exception):
        ops.gemm_nt(A.cpu(), A.cpu())


def test_fused_adamw_matches_torch():
    from smb_vision_amd.optim import SmbvAdamW
    torch.manual_seed(0)
    shapes = [(300, 77), (5,), (64, 3, 4, 4), (70001,), (1, 1, 96)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2], "weight_decay": 0.05}, {"params": ps[2:], "weight_decay": 0.0, "lr": 3e-3}]
    oa = SmbvAdamW(groups(pa), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, max_grad_norm=1.0)
    ob = torch.optim.AdamW(groups(pb), lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    for it in range(4):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (3.0 if it % 2 == 0 else 0.01)  # exercise both the clipped and the unclipped branch
            a.grad, b.grad = g.clone(), g.clone()
        norm = torch.nn.utils.clip_grad_norm_(pb, 1.0)
        ob.step()
        oa.step()
        assert abs(float(oa.last_grad_norm.sqrt()) - float(norm)) <= 1e-5 * float(norm)
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), it
    sa, sb = oa.state[pa[3]], ob.state[pb[3]]
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12) and float(sa["step"]) == 4


@pytest.mark.parametrize("Dh", [64, 32])
@pytest.mark.parametrize("B,S,H,Sq,qoff", [(2, 200, 2, 120, 80), (1, 300, 1, 45, 255), (1, 512, 2, 384, 128)])
def test_attention_query_subrange(ops, B, S, H, Sq, qoff, Dh):
    """Only queries qoff..qoff+Sq-1 of each sequence are computed (keys/values: all S) - the last decoder block's case."""
    w = H * Dh
    qkv = rnd(B * S, 3 * w, seed=11).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    qf = qkv.float().requires_grad_(True)
    q, k, v, s = _attn_ref(qf, B, S, H, Dh)
    s = s[:, :, qoff:qoff + Sq]
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Sq, w)
    got = O.view(B, S, w)[:, qoff:qoff + Sq]
    assert rel_err(got, ref) < 1.5e-2 and rms_err(got, ref) < 6e-3
    assert float(O.view(B, S, w)[:, :qoff].abs().max()) == 0  # rows outside the range are not written
    assert rel_err(lse, torch.logsumexp(s, -1)) < 2e-3
    dO = torch.zeros(B, S, w, device="cuda")
    dO[:, qoff:qoff + Sq] = rnd(B, Sq, w, seed=12)
    dO = dO.bfloat16().view(B * S, w)
    dqkv, delta = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    ref.backward(dO.float().view(B, S, w)[:, qoff:qoff + Sq])
    for i, name in enumerate("qkv"):
        a, b = dqkv[:, i * w:(i + 1) * w], qf.grad[:, i * w:(i + 1) * w]
        assert rel_err(a, b) < 2.5e-2 and rms_err(a, b) < 1e-2, name


@pytest.mark.parametrize("Dh", [64, 32])
def test_attention_kernels_are_run_to_run_identical(ops, Dh):
    """No atomics anywhere and every LDS hand-over (LDS-DMA ring slots, tile seams) is ordered by counted waits + barriers: repeated launches
    on a shape with many tiles per block must agree bit for bit (a race would show up as run-to-run differences)."""
    B, S, H = 1, 4224, 4   # 33 tiles of 128: ring wrap-around, odd tile count, ragged nothing
    w = H * Dh
    qkv = rnd(B * S, 3 * w, seed=31).bfloat16()
    dO = rnd(B * S, w, seed=32).bfloat16()
    O0, lse0 = ops.attention_fwd(qkv, B, S, H, Dh=Dh)
    d0, _ = ops.attention_bwd(qkv, O0, dO, lse0, B, S, H, Dh=Dh)
    for _ in range(8):
        O, lse = ops.attention_fwd(qkv, B, S, H, Dh=Dh)
        assert torch.equal(O, O0) and torch.equal(lse, lse0)
        d, _ = ops.attention_bwd(qkv, O0, dO, lse0, B, S, H, Dh=Dh)
        assert torch.equal(d, d0)


@pytest.mark.parametrize("M,N,K", [(7168, 2304, 768), (1000, 1152, 384), (20480, 384, 1536), (257, 128, 32), (4096, 4096, 96)])
def test_gemm_nt_ring_kernel_matches_reference_and_first_generation(ops, M, N, K):
    """gemm_nt_ring_kernel (256 x 128 tiles, LDS ring of 32-deep K slices, weight rows permuted in LDS so that the register epilogue writes whole
    lines): every epilogue kind against fp32 PyTorch, and - same MFMA shape, same K order - bit for bit against the 128 x 128 kernel."""
    import ctypes
    from smb_vision_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    A = rnd(M, K, seed=1).bfloat16()
    B = rnd(N, K, seed=2, scale=0.05).bfloat16()
    bias = rnd(N, seed=3)
    resid = rnd(M, N, seed=4)
    pre_b = rnd(M, N, seed=7).bfloat16()
    ref = A.float() @ B.float().t()
    runs = {}
    for v in (131072, 256, 131072 | 512, 131072 | 262144):   # ring (3 slots, two blocks per CU) forced, first generation forced, ring with 4 slots, ring with 64-deep slices
        raw.smbv_debug_set_variant(v)
        try:
            runs[v] = (ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True), ops.gemm_nt(A, B, bias=bias, gelu=True, save_pre=True),
                       ops.gemm_nt(A, B, dgelu_pre=pre_b), ops.gemm_nt(A, B, out_bf16=False, out_f32=True, alpha=0.5))
        finally:
            raw.smbv_debug_set_variant(0)
    r0, r1, r2, r3 = runs[131072]
    assert rel_err(r0["f32"], ref + bias + resid) < 1e-5 * math.sqrt(K) + 1e-5
    assert rel_err(r1["pre"], ref + bias) < 6e-3
    assert rel_err(r1["bf16"], torch.nn.functional.gelu(r1["pre"].float())) < 6e-3
    x = pre_b.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel_err(r2["bf16"], ref * x.grad) < 8e-3
    assert rel_err(r3["f32"], 0.5 * ref) < 1e-5 * math.sqrt(K) + 1e-5
    for v in (256, 131072 | 512, 131072 | 262144):
        for a, b in zip(runs[131072], runs[v]):
            for k in a:
                assert torch.equal(a[k], b[k]), (v, k)


@pytest.mark.parametrize("M,N,K", [(7168, 768, 3072), (7168, 768, 2304), (20480, 384, 1536), (20480, 384, 1152), (7168, 1024, 4096), (6200, 768, 1152)])
def test_gemm_nt_splitk_hand_off_is_exact_and_leaves_no_state(ops, M, N, K):
    """Round 5: the long-K / narrow-N products (fc2 forward, d_fc1 and d_qkv input gradients) as two K halves per 256 x 128 tile, the first half's fp32
    partial handed to the second workgroup through scratch (gemm_nt_ring.hip KS = 2).  Checked against fp32 PyTorch at the ring kernel's tolerance and
    against the unsplit product (same tiles, same MFMA: only the association of the K sum differs - a few fp32 ulps); and, as the guide asks of every
    inter-workgroup hand-off, under REUSE of the scratch with CHANGING data: 24 launches alternating between two operand sets through one workspace, each
    of which must reproduce its own first result bit for bit (a consumer that read a stale partial - last launch's, from a cache - or a flag left set
    would show as the other set's numbers), with the device status word clean at the end.  A ragged last row of tiles (M = 6200) included."""
    from smb_vision_amd import _lib
    sets = []
    for sd in (1, 11):
        A = rnd(M, K, seed=sd).bfloat16()
        B = rnd(N, K, seed=sd + 1, scale=0.05).bfloat16()
        sets.append((A, B, rnd(N, seed=sd + 2), rnd(M, N, seed=sd + 3)))
    firsts, reuse = [], None
    for A, B, bias, resid in sets:
        r = ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True, splitk=True, reuse=None if reuse is None else {"bf16": torch.empty_like(reuse["bf16"]),
                                                                                                              "f32": torch.empty_like(reuse["f32"]), "splitk_ws": reuse["splitk_ws"]})
        reuse = r
        ref = A.float() @ B.float().t() + bias + resid
        assert rel_err(r["f32"], ref) < 1e-5 * math.sqrt(K) + 1e-5
        plain = ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True)
        assert rel_err(r["f32"], plain["f32"]) < 2e-6 and float((r["bf16"].float() - plain["bf16"].float()).abs().max()) <= 2 ** -7 * float(plain["bf16"].float().abs().max())
        firsts.append((r["f32"].clone(), r["bf16"].clone()))
    out = {"bf16": torch.empty_like(reuse["bf16"]), "f32": torch.empty_like(reuse["f32"]), "splitk_ws": reuse["splitk_ws"]}
    for it in range(24):
        A, B, bias, resid = sets[it & 1]
        r = ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True, splitk=True, reuse=out)
        assert torch.equal(r["f32"], firsts[it & 1][0]) and torch.equal(r["bf16"], firsts[it & 1][1]), it
    torch.cuda.synchronize()
    assert ops.device_status() == 0, _lib.last_error()
    assert int(out["splitk_ws"][:4096].max()) == 0                      # every consumer cleared its flag
    # other epilogues of the plain kind: scaled fp32 only, output-row remap
    r = ops.gemm_nt(A, B, out_bf16=False, out_f32=True, alpha=0.5, splitk=True)
    assert rel_err(r["f32"], 0.5 * (A.float() @ B.float().t())) < 1e-5 * math.sqrt(K) + 1e-5
    with pytest.raises(ValueError):
        ops.gemm_nt(A[:512], B, splitk=True)                              # 2 x 6 tiles: not a product this form takes
    with pytest.raises(ValueError):
        ops.gemm_nt(A, B, gelu=True, save_pre=True, splitk=True)          # GELU epilogues stay with the unsplit kernels


@pytest.mark.parametrize("M,N,K", [(7168, 768, 768), (7168, 768, 3072), (13312, 384, 4096), (7000, 768, 320), (6450, 768, 64)])
def test_gemm_nt_mixed_tile_launch_is_bitwise_the_plain_one(ops, M, N, K):
    """Grids of 257..384 tiles are launched with their last tile rows cut into 128 x 64 halves (gemm.hip: at most 1.5 tiles of work on any CU instead of
    2): every epilogue kind must come out bit for bit as from the all-128 x 128 launch (debug bit 1024) - same MFMA per element, same K order -
    including a ragged last row of tiles (M = 7000, 6450)."""
    import ctypes
    from smb_vision_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    assert 256 < -(-M // 128) * (N // 128) <= 384
    A = rnd(M, K, seed=1).bfloat16()
    B = rnd(N, K, seed=2, scale=0.05).bfloat16()
    bias = rnd(N, seed=3)
    resid = rnd(M, N, seed=4)
    pre_b = rnd(M, N, seed=7).bfloat16()
    runs = {}
    for v in (0, 1024):
        raw.smbv_debug_set_variant(v)
        try:
            runs[v] = (ops.gemm_nt(A, B, bias=bias, rowadd=resid, out_bf16=True, out_f32=True), ops.gemm_nt(A, B, bias=bias, gelu=True, save_pre=True),
                       ops.gemm_nt(A, B, dgelu_pre=pre_b), ops.gemm_nt(A, B, out_bf16=False, out_f32=True, alpha=0.5))
        finally:
            raw.smbv_debug_set_variant(0)
    ref = A.float() @ B.float().t()
    assert rel_err(runs[0][0]["f32"], ref + bias + resid) < 1e-5 * math.sqrt(K) + 1e-5
    for a, b in zip(runs[0], runs[1024]):
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("M,N1,N2", [(300, 136, 72), (4096, 128, 128), (7168, 768, 3072), (4100, 3072, 768), (2000, 2304, 768), (26, 64, 4096)])
def test_gemm_tn_bias_column_sums_ride_along(ops, M, N1, N2):
    """smbv_gemm_tn_bias: weight gradient AND bias gradient (column sums of dY) of a linear from ONE pass - the sums are an extra X^T * ones MFMA
    product inside the blocks of the first N2 tile (register-staged, LDS-DMA and 256 x 256 kernels, with and without token slabs)."""
    X = rnd(M, N1, seed=11).bfloat16()
    Y = rnd(M, N2, seed=12).bfloat16()
    out, xs = ops.gemm_tn_bias(X, Y)
    assert torch.equal(out, ops.gemm_tn(X, Y))
    ref = X.float().sum(0)
    assert float((xs - ref).abs().max()) <= 2e-6 * M * float(X.float().abs().max()) + 1e-6
    assert rel_err(xs, ref) < 1e-5 * math.sqrt(M) + 1e-6 or float(ref.abs().max()) < 1.0
    out2, xs2 = ops.gemm_tn_bias(X, Y)
    assert torch.equal(xs, xs2) and torch.equal(out, out2)


@pytest.mark.parametrize("Dh,Sq,qoff", [(64, 0, 0), (32, 0, 0), (64, 300, 200), (32, 77, 500)])
def test_attention_residual_output_feeds_delta(ops, Dh, Sq, qoff):
    """ABI 3000: the forward's optional second output O_lo = bf16(O_fp32 - O) and the backward's delta = rowsum(dO * (O + O_lo)), through both delta
    paths the library has (attn_delta_kernel here; the out-projection epilogue is covered by the step tests).  O and LSE do not depend on the option."""
    B, S, H = 2, 777, 3
    w = H * Dh
    qkv = rnd(B * S, 3 * w, seed=11).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    O2, lse2, O_lo = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, resid=True)
    assert torch.equal(O, O2) and torch.equal(lse, lse2)
    n = Sq if Sq > 0 else S
    rows = slice(qoff, qoff + n)
    q, k, v, s = _attn_ref(qkv, B, S, H, Dh)
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, S, w)[:, rows]
    hi, lo = O.float().view(B, S, w)[:, rows], O_lo.float().view(B, S, w)[:, rows]
    assert float(lo.abs().max()) > 0 and bool((lo.abs() <= hi.abs() * 2.0 ** -8 + 1e-30).all())          # half a bf16 ulp of O at most
    assert rms_err(hi + lo, ref) < 0.95 * rms_err(hi, ref)     # it carries the fp32 output's low bits (what is left is the bf16 rounding of P, not of O)
    outside = torch.ones(S, dtype=torch.bool); outside[rows] = False
    assert float(O_lo.float().view(B, S, w)[:, outside].abs().max() if outside.any() else 0.0) == 0.0   # rows outside the query range are not written
    dO = torch.zeros(B, S, w, device="cuda")
    dO[:, rows] = rnd(B, n, w, seed=12)
    dO = dO.bfloat16().view(B * S, w)
    _, delta = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, O_lo=O_lo)
    _, delta0 = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    dv = dO.float().view(B, S, H, Dh)[:, rows]
    want = (dv * (hi + lo).view(B, n, H, Dh)).sum(-1).transpose(1, 2)          # [B, H, n]
    want0 = (dv * hi.view(B, n, H, Dh)).sum(-1).transpose(1, 2)
    assert rel_err(delta, want) < 1e-5 and rel_err(delta0, want0) < 1e-5
    assert not torch.equal(delta, delta0)


def _bwd_case(ops, B, S, H, Dh, Sq, qoff, seed):
    w = H * Dh
    qkv = rnd(B * S, 3 * w, seed=seed).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    n = Sq if Sq > 0 else S
    dO = torch.zeros(B, S, w, device="cuda")
    dO[:, qoff:qoff + n] = rnd(B, n, w, seed=seed + 1)
    return qkv, O, lse, dO.bfloat16().view(B * S, w)


@pytest.mark.parametrize("B,S,H,Dh,Sq,qoff", [(1, 1000, 3, 64, 0, 0), (2, 777, 2, 64, 300, 200), (1, 2176, 2, 32, 181, 1056), (2, 640, 3, 32, 0, 0)])
def test_attention_backward_launch_forms_are_bitwise_equivalent(ops, B, S, H, Dh, Sq, qoff):
    """the shipped ONE-launch pair with mixed block roles, the two-launch path (debug bit 524288: also what a ninth stream of a device gets) and
    the query-owner block on the 32x32x16 MFMA (bit 2) run the same arithmetic in the same order per tile - which block computes which tile, and in how
    many launches, must not change a bit.  (16x16x32 vs 32x32x16 sum a tile's products in a different order: equal to rounding, not bitwise.)"""
    from smb_vision_amd import _lib
    lib = _lib.load()
    qkv, O, lse, dO = _bwd_case(ops, B, S, H, Dh, Sq, qoff, seed=21)
    try:
        base, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        lib.smbv_debug_set_variant(524288)
        two, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        lib.smbv_debug_set_variant(2)
        q32, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        lib.smbv_debug_set_variant(2 | 524288)
        q32_two, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    finally:
        lib.smbv_debug_set_variant(0)
    assert torch.equal(base, two)
    assert torch.equal(q32, q32_two)
    w = H * Dh
    assert torch.equal(base[:, w:], q32[:, w:])                      # dK, dV: the key-owner block is the same in all four
    assert rms_err(base[:, :w], q32[:, :w]) < 2e-3                   # dQ: two MFMA shapes, same products


def test_attention_backward_on_two_streams_of_one_device(ops):
    """The pair launch's scheduling slot belongs to the stream (attention_bwd.hip pair_slot_for): two streams of one device that run attention backwards at
    the same time (the exported smbv_attention_bwd under user streams, two models) use different slots and get the single-stream results bit for bit."""
    qkv, O, lse, dO = _bwd_case(ops, 1, 4096, 4, 64, 0, 0, seed=31)
    want, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 4096, 4)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                outs.append(ops.attention_bwd(qkv, O, dO, lse, 1, 4096, 4)[0])
    torch.cuda.synchronize()
    assert ops.device_status() == 0
    for o in outs:
        assert torch.equal(o, want)


def test_attention_backward_scheduling_fault_is_an_error_not_wrong_gradients(ops):
    """the pair launch keeps its tile queues in device memory the library owns.  A slot that is not in the state its launch expects used
    to give silently missing gradient tiles.  Now every word carries its launch's epoch and a block that finds both queues exhausted reports it: the
    device's sticky status turns SMBV_ERR_STATE and every later step / attention call is refused until it is cleared."""
    from smb_vision_amd import _lib
    qkv, O, lse, dO = _bwd_case(ops, 1, 1024, 2, 64, 0, 0, seed=41)
    want, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
    torch.cuda.synchronize()
    assert ops.device_status() == 0
    try:
        ops.debug_pair_state(1)                                       # every slot's key-owner queue starts 3 tiles in: 3 tiles of the next launch are skipped
        bad, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
        torch.cuda.synchronize()
        assert not torch.equal(bad, want)                             # (the old behaviour's symptom: this result would have been returned as good)
        assert ops.device_status() == -5
        assert "0x5002" in _lib.last_error() or "0x5001" in _lib.last_error()
        with pytest.raises(_lib.SmbvError, match="device fault"):
            ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
    finally:
        ops.debug_pair_state(0)
        torch.cuda.synchronize()
        assert ops.device_status(clear=True) in (0, -5)
    assert ops.device_status() == 0
    again, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
    torch.cuda.synchronize()
    assert torch.equal(again, want) and ops.device_status() == 0


def test_hand_off_words_of_a_faulted_or_unfinished_split_launch_are_never_taken_for_a_later_launchs(ops):
    """the split sweeps of the attention backward and the split grouped weight-gradient launch hand fp32 partial sums between workgroups
    through flags the LIBRARY keeps in device memory.  A launch that faults or runs short leaves flags raised; a later launch's consumer that took such a word for
    its own producer's would add stale or half-written scratch - silently.  Now a flag carries its launch's epoch (attention) / sequence number (weight gradients),
    the weight-gradient launch claims its units from a ticket inside a run the host reserved, and smbv_debug_pair_state(0) restores all of it.  Pinned here:
    (a) flags raised as an EARLIER launch would have left them change nothing - bitwise; (b) a launch into dirtied state is a device fault, not a result;
    (c) reset + clear, and the next split launches reproduce the first bit for bit; (d) a second stream has a slot of its own."""
    from smb_vision_amd import _lib
    lib = _lib.load()
    B, S, H = 1, 7168, 12
    qkv, O, lse, dO = _bwd_case(ops, B, S, H, 64, 0, 0, seed=43)
    aws = torch.empty(lib.smbv_attention_bwd_split_workspace_bytes(B, S, H, 64), dtype=torch.uint8, device="cuda")
    shapes = [(1024, 768, 512), (1024, 512, 768), (1024, 256, 256)]
    pairs = [(rnd(M, N1, seed=30 + i).bfloat16(), rnd(M, N2, seed=40 + i).bfloat16()) for i, (M, N1, N2) in enumerate(shapes)]
    nseg = [3, 1, 1]

    def attn():
        return ops.attention_bwd(qkv, O, dO, lse, B, S, H, split=aws)[0]

    def tn():
        outs, xs = ops.gemm_tn_grouped(pairs, sums=nseg, split=True)
        return torch.cat([o.flatten() for o in outs] + [t for x in xs for t in x])

    torch.cuda.synchronize()
    assert ops.device_status() == 0, _lib.last_error()
    try:
        ops.debug_pair_state(0)                       # epochs / sequence numbers restart: the launches below carry 0, 1, 2 ... and 1, 2, 3 ...
        a0, t0 = attn(), tn()                         # attention epoch 0 (flags raised to 1), weight gradients sequence number 1
        a1, t1 = attn(), tn()
        assert torch.equal(a0, a1) and torch.equal(t0, t1)
        ops.debug_pair_state(2 + 4 * 0)               # (a) every flag of every slot = 1: what epoch 0 / launch 1 would have left had its consumers never run
        a2, t2 = attn(), tn()                         # epoch 2 waits for 3, launch 3 for 3: the stale 1 is overwritten by the producer, never consumed
        torch.cuda.synchronize()
        assert ops.device_status() == 0, _lib.last_error()
        assert torch.equal(a2, a0) and torch.equal(t2, t0)
        ops.debug_pair_state(1)                       # (b) tile queue + 3 / ticket + 3
        bad_t = tn()                                  # three producer units are never claimed: their consumers give up after the bounded wait (~1 s), three blocks draw tickets past the run
        torch.cuda.synchronize()
        assert ops.device_status() == -5
        assert "0x5004" in _lib.last_error() or "0x5005" in _lib.last_error()
        assert not torch.equal(bad_t, t0)
        assert ops.device_status(clear=True) == -5
        bad_a = attn()                                # its last three consumer halves are never claimed: flags stay raised with this launch's epoch
        torch.cuda.synchronize()
        assert ops.device_status() == -5 and ("0x5002" in _lib.last_error() or "0x5001" in _lib.last_error())
        assert not torch.equal(bad_a, a0)
    finally:
        ops.debug_pair_state(0)                       # (c)
        torch.cuda.synchronize()
        ops.device_status(clear=True)
    assert ops.device_status() == 0
    for _ in range(2):
        assert torch.equal(attn(), a0) and torch.equal(tn(), t0)
    side = torch.cuda.Stream()                        # (d)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ts, as_ = tn(), attn()
    side.synchronize()
    assert torch.equal(ts, t0) and torch.equal(as_, a0)
    assert torch.equal(tn(), t0)
    torch.cuda.synchronize()
    assert ops.device_status() == 0, _lib.last_error()


@pytest.mark.parametrize("B,S,H,Dh,Sq,qoff", [(1, 7168, 12, 64, 0, 0), (1, 20480, 6, 64, 13312, 7168), (2, 4096, 6, 64, 0, 0), (1, 7168, 12, 32, 0, 0), (1, 6100, 8, 64, 0, 0)])
def test_attention_backward_split_sweeps_hand_off_exactly(ops, B, S, H, Dh, Sq, qoff):
    """Round 5: smbv_attention_bwd_split - the pair launch's last 192 tiles of each role as two half sweeps whose fp32 accumulators are handed from one workgroup to
    the next through scratch (attention_bwd.hip SweepPart).  Against the unsplit launch the split tiles' sums only associate differently before the one bf16
    rounding: every gradient within a bf16 ulp of its row's scale, and NOT a single element different outside the split tiles' rows would be too strong a claim
    to test cheaply - the norm bound is what is pinned.  And, as for every inter-workgroup hand-off: 16 launches alternating two operand sets through ONE scratch
    buffer, each reproducing its own first result bit for bit (a stale partial or a flag left set would show as the other set's numbers), the status word clean.
    Shapes: the encoder grid, the ranged decoder block (13,312 of 20,480 queries), batch 2, head_dim 32, a ragged last tile."""
    from smb_vision_amd import _lib
    lib = _lib.load()
    need = lib.smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh)
    assert need > 0                                                # these grids are large enough to split (384+ tiles per role)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    sets = [_bwd_case(ops, B, S, H, Dh, Sq, qoff, seed=sd) for sd in (61, 71)]
    firsts = []
    for qkv, O, lse, dO in sets:
        whole, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
        split, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws)
        torch.cuda.synchronize()
        assert ops.device_status() == 0, _lib.last_error()
        d = (split.float() - whole.float())
        assert float(d.norm()) <= 2e-3 * float(whole.float().norm())                      # measured ~3e-4: bf16 re-rounding of sums that differ in their last fp32 bits
        assert float(d.abs().max()) <= 2.0 ** -6 * float(whole.float().abs().max())
        firsts.append(split.clone())
    out = (torch.empty_like(firsts[0]), torch.empty(B, H, Sq if Sq > 0 else S, dtype=torch.float32, device="cuda"))
    for it in range(16):
        qkv, O, lse, dO = sets[it & 1]
        if Sq > 0:
            out[0].zero_()                                                                # (rows outside the query range carry no dQ: the op leaves them as it finds them)
        got, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws, out=out)
        assert torch.equal(got, firsts[it & 1]), it
    torch.cuda.synchronize()
    assert ops.device_status() == 0, _lib.last_error()
    with pytest.raises(_lib.SmbvError, match="rc=-4"):
        ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws[: need - 256])
    if Dh == 64:   # the fp8 query-owner blocks' half sweeps (config 5): dK / dV are the bf16 blocks' and stay bit-identical to the bf16 split launch, dQ within e4m3 noise of it
        f8_whole, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, fp8=True)
        f8_split, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, fp8=True, split=ws)
        again, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, fp8=True, split=ws)
        torch.cuda.synchronize()
        assert ops.device_status() == 0, _lib.last_error()
        w = H * Dh
        assert torch.equal(f8_split, again) and torch.equal(f8_split[:, w:], firsts[1][:, w:])
        dq = f8_split[:, :w].float() - f8_whole[:, :w].float()
        assert float(dq.norm()) <= 2e-3 * float(f8_whole[:, :w].float().norm())


@pytest.mark.parametrize("B,S,H,Dh,real,Sq,qoff,with_pos", [
    (1, 7168, 12, 64, 64, 0, 0, False),       # large enough for half sweeps: the consumer halves store through the map too
    (2, 1000, 3, 64, 64, 0, 0, True),         # batch 2, ragged tiles, shuffled position ids
    (1, 2048, 6, 32, 32, 0, 0, True),         # head_dim 32: seg = 10, a 4-column group straddles two axes
    (1, 1536, 4, 64, 40, 0, 0, True),         # head_dim 40 in a 64-wide slot (seg = 12: columns 36.. pass through)
    (1, 4096, 4, 64, 64, 1024, 3072, True),   # ranged queries: dQ rows outside the range stay as they were
])
def test_attention_backward_stores_through_the_transposed_rotary_map_bitwise(ops, B, S, H, Dh, real, Sq, qoff, with_pos):
    """smbv_attention_bwd_rope (ABI 3004; modeling_vjepa.py:311-326's rotation in the backward): dQ and dK of the rotated q / k leave through the transposed
    3-axis map inside the kernel's stores - bitwise the matrix that smbv_attention_bwd(_split) followed by smbv_rope3d(transpose) leaves, which
    tests/test_vjepa_gpu.py::test_rope3d_matches_oracle pins against autograd of the oracle's map.  Also with the fp8 query-owner blocks, which keep the launch
    behind the kernels, and run to run."""
    from smb_vision_amd import _lib
    from smb_vision_amd.modeling_vjepa import rope_table
    lib = _lib.load()
    grid = 8
    w = H * Dh
    qkv, O, lse, dO = _bwd_case(ops, B, S, H, Dh, Sq, qoff, seed=83)
    g = torch.Generator().manual_seed(5)
    pos = torch.randint(0, grid * grid * 8, (B * S,), generator=g).to(torch.int32).cuda() if with_pos else None
    rope = dict(table=rope_table(real, 128 if not with_pos else 16, "cuda"), pos=pos, S=S, grid=grid, Dh=real, head_stride=Dh)
    need = lib.smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh)
    for split in ([False, True] if need > 0 else [False]):
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device="cuda") if split else None
        two, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws)
        plain = two.clone()
        ops.rope3d_(two, w, H, rope, transpose=True)
        fused, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws, rope=rope)
        again, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws, rope=rope)
        torch.cuda.synchronize()
        assert ops.device_status() == 0, _lib.last_error()
        assert torch.equal(fused, two) and torch.equal(fused, again)
        assert not torch.equal(fused[:, :2 * w], plain[:, :2 * w]) and torch.equal(fused[:, 2 * w:], plain[:, 2 * w:])   # dQ | dK mapped, dV untouched
    unknown, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, split=ws, rope=dict(rope, table_rows=0))   # extent unknown: the launch behind the kernels
    assert torch.equal(unknown, two)
    if Dh == 64 and real == 64:
        two, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, fp8=True)
        ops.rope3d_(two, w, H, rope, transpose=True)
        fused, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, fp8=True, rope=rope)
        torch.cuda.synchronize()
        assert torch.equal(fused, two)
    with pytest.raises(_lib.SmbvError, match="rc=-3"):   # dK / dV only: the map needs both gradient products of the launch
        ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh, parts=3, rope=rope)


def test_attention_backward_argument_error_does_not_poison_the_stream(ops):
    """an attention backward that returns an error AFTER its stream's scheduling slot was looked up - here smbv_attention_bwd_fp8 with a
    workspace one byte short: SMBV_ERR_WORKSPACE from the operand pre-pass - must not consume the slot's launch epoch: the next (valid) launch on the same
    stream finds the device slot stamped with the epoch it carries, gives the right gradients and leaves the status word clean.  (Before: host and device
    were one epoch apart from then on, every later pair launch on that stream ran no tile and raised 0x5001 - a recoverable argument error had become a
    permanent SMBV_ERR_STATE.)"""
    from smb_vision_amd import _lib
    qkv, O, lse, dO = _bwd_case(ops, 1, 1024, 2, 64, 0, 0, seed=43)
    want, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
    torch.cuda.synchronize()
    assert ops.device_status() == 0
    for _ in range(3):   # several failures in a row: each one must leave the epoch where it was
        with pytest.raises(_lib.SmbvError, match=r"rc=-4"):
            ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2, fp8=True, fp8_ws_short=1)
    again, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2)
    torch.cuda.synchronize()
    assert ops.device_status() == 0, _lib.last_error()
    assert torch.equal(again, want)
    f8, _ = ops.attention_bwd(qkv, O, dO, lse, 1, 1024, 2, fp8=True)     # and the fp8 form itself still runs on that stream
    torch.cuda.synchronize()
    assert ops.device_status() == 0 and torch.isfinite(f8.float()).all()
    w = 2 * 64
    assert torch.equal(f8[:, w:], want[:, w:])                             # dK, dV are the bf16 blocks' (bit-identical by construction)


def _attn_bwd_emulated(qkv, O, dO, lse, B, S, H, Dh, Sq, qoff, rounding=True, O_lo=None):
    """fp32 torch restatement of attn_dkv2_kernel / attn_dq2_kernel WITH their bf16 rounding points (attention_bwd.hip): the key-owner kernel scores
    bf16(K * c) against Q, the query-owner kernel bf16(Q * c) against K (c = scale * log2 e), P = exp2(s' - lse * log2 e) is rounded to bf16 for the
    dV product only, dS = P_fp32 * (dP - delta) is rounded to bf16 for the dK / dQ products, outputs are rounded to bf16.  rounding=False: the same
    formulas in exact fp32 (no autograd graph: one S x S matrix per (batch, head) at a time, so the full-size grids fit)."""
    w = H * Dh
    c = Dh ** -0.5 * 1.4426950408889634
    bf = (lambda t: t.bfloat16().float()) if rounding else (lambda t: t)
    f = qkv.float().view(B, S, 3, H, Dh)
    dOv, Ov = dO.float().view(B, S, H, Dh), O.float().view(B, S, H, Dh)
    if O_lo is not None:   # delta from O + O_lo (the forward's bf16 residual of its fp32 output)
        Ov = Ov + O_lo.float().view(B, S, H, Dh)
    out = [torch.zeros(B, S, H, Dh, device=qkv.device) for _ in range(3)]
    for b in range(B):
        for h in range(H):
            q, k, v = f[b, :, 0, h], f[b, :, 1, h], f[b, :, 2, h]                                       # [S, Dh]
            qs, do, o = q[qoff:qoff + Sq], dOv[b, qoff:qoff + Sq, h], Ov[b, qoff:qoff + Sq, h]
            l2 = (lse[b, h] * 1.4426950408889634).unsqueeze(-1)                                          # [Sq, 1]
            if not rounding:  # exact arithmetic throughout: the row statistics and the attention output too (the kernels take delta = rowsum(dO * O) from
                sx = qs @ k.t() * Dh ** -0.5                                                             # the bf16 O the forward stored, as flash attention does)
                l2 = torch.logsumexp(sx, -1, keepdim=True) * 1.4426950408889634
                o = torch.softmax(sx, -1) @ v
                del sx
            delta = (do * o).sum(-1, keepdim=True)
            dp = do @ v.t() - delta
            p = torch.exp2(qs @ bf(k * c).t() - l2)                                                     # key-owner kernel: dK, dV
            out[2][b, :, h] = bf(p).t() @ do
            out[1][b, :, h] = bf(p * dp).t() @ qs * Dh ** -0.5
            p = torch.exp2(bf(qs * c) @ k.t() - l2)                                                     # query-owner kernel: dQ
            out[0][b, qoff:qoff + Sq, h] = bf(p * dp) @ k * Dh ** -0.5
            del p, dp
    return [bf(t).reshape(B * S, w) for t in out]


def test_attention_bwd_dh32_subrange_fuzz_case_is_operand_rounding(ops):
    """tools/fuzz_ops.py (seed 11, case 36: Dh 32, B 3, S 2176, H 2, queries 1056..1236, unit-variance operands x 2 - logits of
    standard deviation ~4, a sharp softmax) reported max|dK - fp32 reference| = 4.1e-2 of max|dK|, above the 4e-2 bound of the time.  This test
    rebuilds that case and compares the kernels with (a) the exact fp32 reference and (b) a reference with the kernels' bf16 rounding points
    (P rounded for dV, dS rounded for dK / dQ, the pre-scaled K / Q operands): if the deviation is rounding, (b) must sit an order of magnitude closer
    than (a) - a tile-seam or Sq / q_off masking error in the Dh = 32 ranged dK / dV path would be O(1) in single rows against BOTH references."""
    B, S, H, Sq, qoff, Dh = 3, 2176, 2, 181, 1056, 32
    w = H * Dh
    g = torch.Generator(device="cuda").manual_seed(36)
    qkv = (torch.randn(B * S, 3 * w, generator=g, device="cuda") * 2.0).bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    dO = torch.zeros(B, S, w, device="cuda")
    dO[:, qoff:qoff + Sq] = torch.randn(B, Sq, w, generator=g, device="cuda")
    dO = dO.bfloat16().view(B * S, w)
    dqkv, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Sq=Sq, q_off=qoff, Dh=Dh)
    qf = qkv.float().requires_grad_(True)
    q, k, v, s = _attn_ref(qf, B, S, H, Dh)
    s = s[:, :, qoff:qoff + Sq]
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Sq, w)
    ref.backward(dO.float().view(B, S, w)[:, qoff:qoff + Sq])
    emu = _attn_bwd_emulated(qkv, O, dO, lse, B, S, H, Dh, Sq, qoff)
    rows = []
    for i, name in enumerate("qkv"):
        got, exact = dqkv[:, i * w:(i + 1) * w], qf.grad[:, i * w:(i + 1) * w]
        rows.append((name, rel_err(got, exact), rms_err(got, exact), rel_err(got, emu[i]), rms_err(got, emu[i])))
    print("Dh32 ranged backward, fuzz case 11/36 (max-abs, rms vs exact fp32 | vs the same rounding points): " + "; ".join(f"d{n} {a:.1e} {b:.1e} | {c:.1e} {d:.1e}" for n, a, b, c, d in rows))
    for name, mx_exact, rms_exact, mx_emu, rms_emu in rows:
        assert mx_exact < 5e-2 and rms_exact < 1.5e-2, (name, mx_exact, rms_exact)          # the fuzz tool's (measured) bound
        # same rounding points: what remains is fp32 summation order and output roundings that fall the other way (measured: max-abs 5.5e-5 .. 7.9e-4,
        # rms 3.6e-6 .. 1.0e-4 - two orders of magnitude below the distance to exact arithmetic: the fuzz tool's 4.1e-2 was operand rounding)
        assert mx_emu < 4e-3 and rms_emu < 1e-3, (name, mx_emu, rms_emu)


def test_attention_bwd_cancellation_regime_at_the_vjepa_encoder_grid(ops):
    """The V-JEPA encoder's attention grid (S = 20,480, 12 heads of 64: the (160, 12)-block backward launches) in the regime its init-scale weights put it
    in: keys / queries = a component common to all tokens + a small token-dependent part, i.e. near-uniform attention.  There dQ = sum_k dS[q,k] K[k]
    cancels almost completely in exact arithmetic (sum_k dS[q,k] = 0: what survives is 1/sqrt(S) of the terms' size), while delta = rowsum(dO * O) is formed
    from the bf16 attention output the forward stored - as in flash attention, the reference's own bf16 path - and its 2^-9 rounding enters dQ multiplied by
    the softmax-weighted MEAN key, which does not cancel: relative to the gradient that is 2^-9 x sqrt(S) x |mean key| / (key spread), i.e. tens of percent at
    S = 20,480.  The kernels sit that far from the exact fp32 gradient and within 1e-3 of the same formulas with their rounding points.  This is what bounds the encoder query / key gradients of tests/test_fullsize_gpu.py's V-JEPA case."""
    B, S, H, Dh = 1, 20480, 12, 64
    w = H * Dh
    g = torch.Generator(device="cuda").manual_seed(5)
    common = torch.randn(1, 3 * w, generator=g, device="cuda")
    qkv = torch.randn(B * S, 3 * w, generator=g, device="cuda") * 0.5 + common   # |mean| / spread = 2, about what the i.i.d.-voxel volumes give the model
    # (values too: the attention output is then ~ the mean value row, and delta = rowsum(dO * O) a sizeable number whose bf16-O rounding matters)
    qkv = qkv.bfloat16()
    dO = torch.randn(B * S, w, generator=g, device="cuda").bfloat16()
    O, lse = ops.attention_fwd(qkv, B, S, H, Dh=Dh)
    dqkv, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Dh=Dh)
    emu = _attn_bwd_emulated(qkv, O, dO, lse, B, S, H, Dh, S, 0)
    exact = _attn_bwd_emulated(qkv, O, dO, lse, B, S, H, Dh, S, 0, rounding=False)
    l2 = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    rows = [(n, l2(dqkv[:, i * w:(i + 1) * w], exact[i]), l2(dqkv[:, i * w:(i + 1) * w], emu[i]), l2(emu[i], exact[i])) for i, n in enumerate("qkv")]
    print("near-uniform attention at S 20480 x 12 heads (relative L2: kernel vs exact | kernel vs same rounding points | emulation vs exact): "
          + "; ".join(f"d{n} {a:.2e} | {b:.2e} | {c:.2e}" for n, a, b, c in rows))
    for n, vs_exact, vs_emu, emu_exact in rows:
        assert vs_emu < 5e-3, (n, vs_emu)                       # the kernels ARE the formulas with those rounding points
        assert vs_exact < 1.5 * emu_exact + 5e-3, (n, vs_exact, emu_exact)   # ... and nothing beyond them separates the kernels from exact arithmetic
    # Round 4: the rounding point removed.  The forward also stores O_lo = bf16(O_fp32 - O) and delta is formed from O + O_lo (16 significant bits):
    # the same launches then sit within the ordinary bf16-operand distance of exact arithmetic in this regime too.
    O2, lse2, O_lo = ops.attention_fwd(qkv, B, S, H, Dh=Dh, resid=True)
    assert torch.equal(O2, O) and torch.equal(lse2, lse)
    assert float(O_lo.float().abs().max()) <= 2.0 ** -8 * float(O.float().abs().max())     # a residual, half an ulp of O at most
    dqkv2, _ = ops.attention_bwd(qkv, O, dO, lse, B, S, H, Dh=Dh, O_lo=O_lo)
    emu2 = _attn_bwd_emulated(qkv, O, dO, lse, B, S, H, Dh, S, 0, O_lo=O_lo)
    rows2 = [(n, l2(dqkv2[:, i * w:(i + 1) * w], exact[i]), l2(dqkv2[:, i * w:(i + 1) * w], emu2[i])) for i, n in enumerate("qkv")]
    print("  with the residual output (kernel vs exact | kernel vs same rounding points): " + "; ".join(f"d{n} {a:.2e} | {b:.2e}" for n, a, b in rows2))
    for n, vs_exact, vs_emu in rows2:
        assert vs_emu < 5e-3, (n, vs_emu)
        assert vs_exact < 5e-2, (n, vs_exact)                   # dQ (and dK) vs exact arithmetic within the step's gradient bound
    assert torch.equal(dqkv2[:, 2 * w:], dqkv[:, 2 * w:])       # dV does not read delta


def test_gemm_operands_beyond_2_gib_take_the_64_bit_paths(ops):
    """The LDS-DMA kernels address their operands with 32-bit byte offsets and are selected only below 2 GiB per operand (gemm.hip, gemm_nt_ring.hip:
    `M * ld * 2 < 2^31`); beyond it the register-staged kernels run (64-bit pointers).  A 2.15-GB activation matrix (350,208 rows x 3,072: B = 17 of the
    bench's decoder MLP) through both product families and the grouped launch's refusal: results against fp32 torch, rows at the far end included."""
    M, K, N = 350_208, 3072, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    for i in range(0, M, 43_776):   # (filled in pieces: no 4.3-GB fp32 temporary)
        A[i:i + 43_776] = torch.randn(43_776, K, generator=g, device="cuda").bfloat16()
    assert A.numel() * 2 >= 1 << 31
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.02).bfloat16()
    out = ops.gemm_nt(A, W)["bf16"]
    for lo in (0, M // 2, M - 4096):   # first rows, middle, and the rows whose byte offsets exceed 2^31
        ref = A[lo:lo + 4096].float() @ W.float().t()
        assert rel_err(out[lo:lo + 4096], ref) < 1e-2, lo
    Y = torch.randn(M, 128, generator=g, device="cuda").bfloat16()
    got = ops.gemm_tn(A, Y)
    ref = torch.zeros(K, 128, dtype=torch.float32, device="cuda")
    for i in range(0, M, 43_776):
        ref += A[i:i + 43_776].float().t() @ Y[i:i + 43_776].float()
    assert rel_err(got, ref) < 1e-3
    with pytest.raises(ValueError, match="not groupable"):   # the grouped launch refuses (the step then issues the products one by one)
        ops.gemm_tn_grouped([(A, Y)])
    assert ops.device_status() == 0
