"""Per-op Python wrappers over the C ABI (torch tensors in, torch tensors out; all work happens in libsmbv.so on the
current HIP stream).  These are what the GPU parity tests call; the model itself drives the fused whole-step entry
points (see modeling_videomae.py).  Nothing here computes with PyTorch — tensors are only device memory."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import GemmEpi, check


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SmbvError("smb_vision_amd ops need tensors on a HIP device (no CPU fallback)")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def gemm_nt(A, B, *, bias=None, rowadd=None, rowidx=None, gelu=False, save_pre=False, dgelu_pre=None, out_bf16=True, out_f32=False,
            alpha=1.0, grp=None, out_rows=None, reuse=None, splitk=False, rope=None):
    """C = epi(A[M,K] @ B[N,K]^T).  Returns dict with 'bf16', 'f32', 'pre' entries as requested.  reuse: a dict returned by an earlier call of the
    same shape - its tensors are written again instead of allocating (and zero-filling) new ones (timing loops).  splitk: smbv_gemm_nt_splitk (two K
    halves per 256 x 128 tile; raises ValueError for a product that form does not take).  rope: dict(table, pos (or None), S, grid, Dh, head_stride, wa) -
    smbv_gemm_nt_rope, the 3-axis rotary embedding of the q | k sections applied in the epilogue."""
    _need_gpu(A, B)
    lib = _lib.load()
    M, K = A.shape
    N = B.shape[0]
    rows = out_rows or M
    res = {}
    e = GemmEpi()
    e.alpha = alpha
    e.bias = _p(bias)
    e.rowadd, e.rowidx, e.ldr = _p(rowadd), _p(rowidx), (rowadd.stride(0) if rowadd is not None else 0)
    e.dgelu_pre, e.ld_pre = _p(dgelu_pre), (dgelu_pre.stride(0) if dgelu_pre is not None else 0)
    e.act_gelu = int(gelu)
    if save_pre:
        res["pre"] = reuse["pre"] if reuse else torch.empty(rows, N, dtype=torch.bfloat16, device=A.device)
        e.aux_pre, e.ld_aux = res["pre"].data_ptr(), N
    if out_bf16:
        res["bf16"] = reuse["bf16"] if reuse else torch.zeros(rows, N, dtype=torch.bfloat16, device=A.device)
        e.out_bf16, e.ld_ob = res["bf16"].data_ptr(), N
    if out_f32:
        res["f32"] = reuse["f32"] if reuse else torch.zeros(rows, N, dtype=torch.float32, device=A.device)
        e.out_f32, e.ld_of = res["f32"].data_ptr(), N
    if grp:
        e.grp_in, e.grp_out = grp
    if splitk:
        ws = (reuse or {}).get("splitk_ws")
        if ws is None:
            ws = _ws(lib.smbv_gemm_nt_splitk_workspace_bytes(M, N), A.device)
        res["splitk_ws"] = ws
        check(lib.smbv_gemm_nt_splitk(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, C.byref(e), _p(ws), ws.numel(), _stream()), "smbv_gemm_nt_splitk")
        return res
    if rope is not None:
        rc = rope_cfg(**{k: v for k, v in rope.items() if k != "wa"})
        check(lib.smbv_gemm_nt_rope(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, C.byref(e), C.byref(rc), rope["wa"], _stream()), "smbv_gemm_nt_rope")
        return res
    check(lib.smbv_gemm_nt(_p(A), A.stride(0), _p(B), B.stride(0), M, N, K, C.byref(e), _stream()), "smbv_gemm_nt")
    return res


def rope_cfg(table, pos, S, grid, Dh, head_stride, table_rows=None):
    """smbv_rope3d_cfg over live tensors (the caller keeps them alive for the call).  table_rows: default the table's extent; 0 = "unknown" (the map then
    runs as the stand-alone launch)."""
    rc = _lib.Rope3dCfg()
    rc.table, rc.pos, rc.S, rc.grid, rc.Dh, rc.head_stride = table.data_ptr(), (pos.data_ptr() if pos is not None else None), S, grid, Dh, head_stride
    rc.table_rows = table.shape[0] if table_rows is None else table_rows
    return rc


def rope3d_(qkv, wa, H, rope, transpose=False):
    """smbv_rope3d in place on the q | k sections of qkv [rows, ld] (rope: the dict gemm_nt takes, wa / table_rows ignored)."""
    lib = _lib.load()
    check(lib.smbv_rope3d(_p(qkv), qkv.stride(0), wa, qkv.shape[0], rope["S"], H, rope["head_stride"], rope["Dh"], rope["grid"],
                          (rope["pos"].data_ptr() if rope["pos"] is not None else None), rope["table"].data_ptr(), int(transpose), _stream()), "smbv_rope3d")
    return qkv


def gemm_tn(X, Y):
    """out[N1,N2] fp32 = X[M,N1]^T @ Y[M,N2]."""
    _need_gpu(X, Y)
    lib = _lib.load()
    M, N1 = X.shape
    N2 = Y.shape[1]
    out = torch.empty(N1, N2, dtype=torch.float32, device=X.device)
    ws = _ws(lib.smbv_gemm_tn_workspace_bytes(M, N1, N2), X.device)
    check(lib.smbv_gemm_tn(_p(X), X.stride(0), _p(Y), Y.stride(0), M, N1, N2, _p(out), _p(ws), ws.numel(), _stream()), "smbv_gemm_tn")
    return out


def gemm_tn_bias(X, Y):
    """(X^T @ Y fp32 [N1,N2], column sums of X fp32 [N1]): weight and bias gradient of a linear in one launch."""
    _need_gpu(X, Y)
    lib = _lib.load()
    M, N1 = X.shape
    N2 = Y.shape[1]
    out = torch.empty(N1, N2, dtype=torch.float32, device=X.device)
    xs = torch.empty(N1, dtype=torch.float32, device=X.device)
    ws = _ws(lib.smbv_gemm_tn_workspace_bytes(M, N1, N2), X.device)
    check(lib.smbv_gemm_tn_bias(_p(X), X.stride(0), _p(Y), Y.stride(0), M, N1, N2, _p(out), _p(xs), _p(ws), ws.numel(), _stream()), "smbv_gemm_tn_bias")
    return out, xs


def gemm_tn_grouped(pairs, sums=None, out=None, split=False):
    """pairs: list of (X [M,N1], Y [M,N2]) bf16; returns the list of X^T @ Y (fp32) from ONE launch (smbv_gemm_tn_grouped).  sums: per job None or a
    segment count (1..3): also returns that job's column sums of X as a list of `segments` tensors.  split: every tile's token range as two halves by two
    workgroups (smbv_gemm_tn_grouped_split)."""
    lib = _lib.load()
    n = len(pairs)
    jobs = (_lib.TnJob * n)()
    outs, xs = [], []
    for i, (X, Y) in enumerate(pairs):
        _need_gpu(X, Y)
        M, N1 = X.shape
        N2 = Y.shape[1]
        o = out[i] if out is not None else torch.empty(N1, N2, dtype=torch.float32, device=X.device)
        outs.append(o)
        j = jobs[i]
        j.X, j.Y, j.out, j.ldx, j.ldy, j.M, j.N1, j.N2 = X.data_ptr(), Y.data_ptr(), o.data_ptr(), X.stride(0), Y.stride(0), M, N1, N2
        nseg = sums[i] if sums else None
        if nseg:
            segs = [torch.empty(N1 // nseg, dtype=torch.float32, device=X.device) for _ in range(nseg)]
            j.seg = N1 // nseg
            for k, t in enumerate(segs):
                j.sum_dst[k] = t.data_ptr()
            xs.append(segs)
        else:
            j.seg = 0
            xs.append(None)
    if split:
        ws = _ws(max(256, lib.smbv_gemm_tn_grouped_split_workspace_bytes(jobs, n)), pairs[0][0].device)
        check(lib.smbv_gemm_tn_grouped_split(jobs, n, _p(ws), ws.numel(), _stream()), "smbv_gemm_tn_grouped_split")
    else:
        check(lib.smbv_gemm_tn_grouped(jobs, n, _stream()), "smbv_gemm_tn_grouped")
    return (outs, xs) if sums else outs


def layernorm_fwd(x, gamma, beta, eps, rows=None, grp=(0, 0, 0)):
    _need_gpu(x, gamma, beta)
    lib = _lib.load()
    D = x.shape[-1]
    rows = rows if rows is not None else x.numel() // D
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib.smbv_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, eps, *grp, _stream()), "smbv_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, dres=None, grp=(0, 0, 0)):
    _need_gpu(dy, x)
    lib = _lib.load()
    rows, D = dy.shape
    dx = torch.zeros_like(x, dtype=torch.float32)
    dxb = torch.zeros_like(x, dtype=torch.bfloat16)
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    db = torch.empty(D, dtype=torch.float32, device=x.device)
    ws = _ws(lib.smbv_layernorm_bwd_workspace_bytes(rows, D), x.device)
    check(lib.smbv_layernorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb), _p(dg), _p(db), rows, D, *grp,
                                 _p(ws), ws.numel(), _stream()), "smbv_layernorm_bwd")
    return dx, dxb, dg, db


def attention_fwd(qkv, B, S, H, scale=None, Sq=0, q_off=0, Dh=64, resid=False):
    """qkv: bf16 [B*S, 3*H*Dh] (q | k | v column blocks), Dh 64 or 32.  Returns (O bf16 [B*S, H*Dh], LSE fp32 [B,H,S]); with ``resid`` also
    O_lo = bf16(O_fp32 - O), which attention_bwd takes to form delta from the output to 16 significant bits."""
    _need_gpu(qkv)
    lib = _lib.load()
    w = H * Dh
    scale = scale if scale is not None else Dh ** -0.5
    O = torch.zeros(B * S, w, dtype=torch.bfloat16, device=qkv.device)
    O_lo = torch.zeros_like(O) if resid else None
    lse = torch.empty(B, H, Sq if Sq > 0 else S, dtype=torch.float32, device=qkv.device)
    base = qkv.data_ptr()
    check(lib.smbv_attention_fwd(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), _p(O_lo), w, _p(lse), B, S, H, Dh, scale, Sq, q_off, _stream()),
          "smbv_attention_fwd")
    return (O, lse, O_lo) if resid else (O, lse)


def attention_fwd_fp8(qkv, B, S, H, scale=None, Sq=0, q_off=0):
    """As attention_fwd, on the block-scaled fp8 (OCP e4m3) MFMA path (smbv_attention_fwd_fp8)."""
    _need_gpu(qkv)
    lib = _lib.load()
    w = H * 64
    scale = scale if scale is not None else 64 ** -0.5
    O = torch.zeros(B * S, w, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, Sq if Sq > 0 else S, dtype=torch.float32, device=qkv.device)
    ws = _ws(lib.smbv_attention_fp8_workspace_bytes(B, S, H), qkv.device)
    base = qkv.data_ptr()
    check(lib.smbv_attention_fwd_fp8(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), w, _p(lse), B, S, H, 64, scale, Sq, q_off, _p(ws), ws.numel(),
                                     _stream()), "smbv_attention_fwd_fp8")
    return O, lse


def attention_bwd(qkv, O, dO, lse, B, S, H, scale=None, parts=7, out=None, Sq=0, q_off=0, Dh=64, O_lo=None, fp8=False, fp8_ws_short=0, split=None, rope=None):
    """rope: dict(table, pos (or None), S, grid, Dh (the real head_dim), head_stride) - smbv_attention_bwd_rope: dQ / dK leave through the transposed rotary map
    (split: None / False = no half sweeps).  fp8: the query-owner blocks recompute S and dP on the block-scaled fp8 MFMA (smbv_attention_bwd_fp8; head_dim 64).  fp8_ws_short: declare the
    workspace that many bytes smaller than it is (the error-path test).  split: True (a fresh scratch) or a uint8 scratch tensor - smbv_attention_bwd_split, the
    form the steps run (the last tiles of each role as two half sweeps)."""
    _need_gpu(qkv, O, dO, lse)
    lib = _lib.load()
    w = H * Dh
    scale = scale if scale is not None else Dh ** -0.5
    dqkv, delta = out if out is not None else (torch.zeros_like(qkv), torch.empty(B, H, Sq if Sq > 0 else S, dtype=torch.float32, device=qkv.device))
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    if rope is not None:
        use_split = split is not None and split is not False
        ws = (split if torch.is_tensor(split) else _ws(max(256, lib.smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh)), qkv.device)) if use_split else None
        f8 = _ws(lib.smbv_attention_bwd_fp8_workspace_bytes(B, S, H), qkv.device) if fp8 else None
        rc = rope_cfg(**{k: v for k, v in rope.items() if k != "wa"})
        check(lib.smbv_attention_bwd_rope(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), _p(O_lo), _p(dO), w, _p(lse), _p(delta), dbase, dbase + 2 * w,
                                          dbase + 4 * w, dqkv.stride(0), B, S, H, Dh, scale, parts, Sq, q_off, _p(ws), ws.numel() if use_split else 0, _p(f8),
                                          f8.numel() if fp8 else 0, C.byref(rc), _stream()), "smbv_attention_bwd_rope")
        return dqkv, delta
    if fp8 and (split is None or split is False):
        ws = _ws(lib.smbv_attention_bwd_fp8_workspace_bytes(B, S, H), qkv.device)
        check(lib.smbv_attention_bwd_fp8(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), _p(O_lo), _p(dO), w, _p(lse), _p(delta), dbase, dbase + 2 * w,
                                         dbase + 4 * w, dqkv.stride(0), B, S, H, Dh, scale, parts, Sq, q_off, _p(ws), ws.numel() - fp8_ws_short, _stream()), "smbv_attention_bwd_fp8")
        return dqkv, delta
    if split is not None and split is not False:
        ws = split if torch.is_tensor(split) else _ws(max(256, lib.smbv_attention_bwd_split_workspace_bytes(B, S, H, Dh)), qkv.device)
        f8 = _ws(lib.smbv_attention_bwd_fp8_workspace_bytes(B, S, H), qkv.device) if fp8 else None
        check(lib.smbv_attention_bwd_split(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), _p(O_lo), _p(dO), w, _p(lse), _p(delta), dbase, dbase + 2 * w,
                                           dbase + 4 * w, dqkv.stride(0), B, S, H, Dh, scale, parts, Sq, q_off, _p(ws), ws.numel(), _p(f8), f8.numel() if fp8 else 0,
                                           _stream()), "smbv_attention_bwd_split")
        return dqkv, delta
    check(lib.smbv_attention_bwd(base, base + 2 * w, base + 4 * w, qkv.stride(0), _p(O), _p(O_lo), _p(dO), w, _p(lse), _p(delta), dbase, dbase + 2 * w,
                                 dbase + 4 * w, dqkv.stride(0), B, S, H, Dh, scale, parts, Sq, q_off, _stream()), "smbv_attention_bwd")
    return dqkv, delta


def gather_bricks(pix, idx, per_batch, ts, ps):
    _need_gpu(pix, idx)
    lib = _lib.load()
    B, T, Cc, H, W = pix.shape
    rows = idx.numel()
    out = torch.empty(rows, ts * ps * ps, dtype=torch.bfloat16, device=pix.device)
    check(lib.smbv_gather_bricks(_p(pix), _p(idx), rows, per_batch, _p(out), T, H, W, ts, ps, _stream()), "smbv_gather_bricks")
    return out


def brick_stats(pix, idx, per_batch, ts, ps):
    _need_gpu(pix, idx)
    lib = _lib.load()
    B, T, Cc, H, W = pix.shape
    rows = idx.numel()
    mean = torch.empty(rows, dtype=torch.float32, device=pix.device)
    rinv = torch.empty(rows, dtype=torch.float32, device=pix.device)
    check(lib.smbv_brick_stats(_p(pix), _p(idx), rows, per_batch, _p(mean), _p(rinv), T, H, W, ts, ps, _stream()), "smbv_brick_stats")
    return mean, rinv


def mse_fwd(pix, idx, per_batch, mean, rinv, logits, ts, ps):
    lib = _lib.load()
    B, T, Cc, H, W = pix.shape
    rows = idx.numel()
    part = torch.empty(rows, dtype=torch.float32, device=pix.device)
    loss = torch.empty(1, dtype=torch.float32, device=pix.device)
    check(lib.smbv_mse_fwd(_p(pix), _p(idx), rows, per_batch, _p(mean), _p(rinv), _p(logits), _p(part), _p(loss), T, H, W, ts, ps, _stream()),
          "smbv_mse_fwd")
    return loss


def mse_bwd(pix, idx, per_batch, mean, rinv, logits, ts, ps, gscale=None):
    lib = _lib.load()
    B, T, Cc, H, W = pix.shape
    rows = idx.numel()
    dl = torch.empty_like(logits)
    check(lib.smbv_mse_bwd(_p(pix), _p(idx), rows, per_batch, _p(mean), _p(rinv), _p(logits), _p(dl), _p(gscale), T, H, W, ts, ps, _stream()),
          "smbv_mse_bwd")
    return dl


def cast_bf16(x):
    lib = _lib.load()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.smbv_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "smbv_cast_f32_bf16")
    return out


def transpose_cast_bf16(x):
    lib = _lib.load()
    R, Cc = x.shape
    out = torch.empty(Cc, R, dtype=torch.bfloat16, device=x.device)
    check(lib.smbv_transpose_cast_f32_bf16(_p(x), _p(out), R, Cc, R, _stream()), "smbv_transpose_cast_f32_bf16")
    return out


def colsum_bf16(X):
    lib = _lib.load()
    M, N = X.shape
    out = torch.empty(N, dtype=torch.float32, device=X.device)
    ws = _ws(lib.smbv_colsum_workspace_bytes(M, N), X.device)
    check(lib.smbv_colsum_bf16(_p(X), X.stride(0), M, N, _p(out), _p(ws), ws.numel(), _stream()), "smbv_colsum_bf16")
    return out


def device_status(clear=False) -> int:
    """smbv_device_status: 0, or SMBV_ERR_STATE (-5) once a kernel of an earlier call has recorded a fault on the current device (text: _lib.last_error())."""
    return int(_lib.load().smbv_device_status(1 if clear else 0))


def debug_pair_state(op: int):
    """smbv_debug_pair_state: 0 = restore the library's device-side scheduling state (attention backward tile queues / tickets / split-sweep flags, the split
    weight-gradient launch's tickets / flags) and the host's launch epochs; test hooks: 1 = dirty the slots, 2 + 4 e = raise every hand-off flag as an unfinished
    launch of epoch e would have left it."""
    check(_lib.load().smbv_debug_pair_state(int(op), _stream()), "smbv_debug_pair_state")

