"""The MI355X attention kernels behind the reference's own operator plug-in point.

Both reference models pick their attention implementation from transformers' registry by name:
`ALL_ATTENTION_FUNCTIONS[config._attn_implementation]` (src/models/vjepa/modeling_vjepa.py:345-364, 790-809, 865-884;
src/models/videomae/modeling_videomae.py:270-289), called as

    attention_interface(module, query[B,H,S,Dh], key, value, mask_or_None, is_causal=..., scaling=..., dropout=...)
        -> (context[B,S,H,Dh], attention_probs_or_None)

and the reference's launch scripts select `flash_attention_2` there (src/run_vjepa.py:230-231).  `register()` adds the name "smbv":
a model built from the reference's unmodified classes runs `smbv_attention_fwd` / `smbv_attention_bwd` (include/smbv.h) for its
softmax(QK^T * scaling) V and everything else in PyTorch.  This is the operator-level drop-in; the fused train step
(modeling_videomae.py / modeling_vjepa.py) is the shipped path and avoids the head-major <-> token-major copies this interface forces.

No fallback: anything the kernels do not compute (masks, dropout, causal, head widths other than 64 / 32, no GPU) raises.
"""
import torch

from . import ops

NAME = "smbv"


class _SmbvAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, value, scaling):
        B, H, S, Dh = query.shape
        # the kernels read token-major rows [B*S, q | k | v column blocks of H*Dh]
        qkv = torch.empty(B * S, 3 * H * Dh, dtype=torch.bfloat16, device=query.device)
        packed = qkv.view(B, S, 3, H, Dh)
        packed[:, :, 0].copy_(query.transpose(1, 2))
        packed[:, :, 1].copy_(key.transpose(1, 2))
        packed[:, :, 2].copy_(value.transpose(1, 2))
        out, lse, out_lo = ops.attention_fwd(qkv, B, S, H, scale=scaling, Dh=Dh, resid=True)  # (+ the bf16 residual of the fp32 output, for delta)
        ctx.save_for_backward(qkv, out, lse, out_lo)
        ctx.dims = (B, H, S, Dh, scaling, query.dtype)
        return out.view(B, S, H, Dh).to(query.dtype)

    @staticmethod
    def backward(ctx, d_out):
        qkv, out, lse, out_lo = ctx.saved_tensors
        B, H, S, Dh, scaling, dtype = ctx.dims
        d_rows = d_out.reshape(B * S, H * Dh).to(torch.bfloat16).contiguous()
        d_qkv, _ = ops.attention_bwd(qkv, out, d_rows, lse, B, S, H, scale=scaling, Dh=Dh, O_lo=out_lo)
        d = d_qkv.view(B, S, 3, H, Dh)
        return (d[:, :, 0].transpose(1, 2).to(dtype), d[:, :, 1].transpose(1, 2).to(dtype), d[:, :, 2].transpose(1, 2).to(dtype), None)


def smbv_attention(module, query, key, value, attention_mask=None, *, scaling=None, dropout=0.0, is_causal=False, **kwargs):
    """transformers attention-interface function (same contract as its `eager_attention_forward`): bf16 MFMA arithmetic, fp32 softmax
    statistics; returns (context [B, S, H, Dh] in the dtype of `query`, None)."""
    if attention_mask is not None:
        raise ValueError("smbv attention: attention / head masks are not supported (the reference's training path passes None)")
    if dropout:
        raise ValueError("smbv attention: attention dropout is not supported (the reference trains with attention_probs_dropout_prob = 0)")
    if is_causal:
        raise ValueError("smbv attention: bidirectional attention only")
    if query.dim() != 4 or key.shape != query.shape or value.shape != query.shape:
        raise ValueError(f"smbv attention: expected query / key / value of one shape [B, H, S, Dh], got {tuple(query.shape)}, "
                         f"{tuple(key.shape)}, {tuple(value.shape)}")
    if query.shape[-1] not in (64, 32):
        raise ValueError(f"smbv attention: head width {query.shape[-1]} has no native build (64 and 32 do)")
    if not query.is_cuda:
        raise RuntimeError("smbv attention needs an MI355X: there is no CPU path")
    if scaling is None:
        scaling = query.shape[-1] ** -0.5
    return _SmbvAttentionFn.apply(query, key, value, float(scaling)), None


def register(name=NAME):
    """Make `config._attn_implementation = name` select the MI355X kernels (transformers.AttentionInterface)."""
    from transformers import AttentionInterface
    AttentionInterface.register(name, smbv_attention)
    return name
