"""CPU oracle for the MIM (VideoMAE-for-pretraining) train step.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch fp32 restatement of the arithmetic of the reference hot path.  It is the
*checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package (``smb_vision_amd``) never does, and fails
loudly when its HIP library is missing.

Specification text followed (reference file:line, relative to /root/reference):

* ``src/models/videomae/modeling_videomae.py:95-106``  get_sinusoid_encoding_table      -> sinusoid_table
* ``src/models/videomae/modeling_videomae.py:124-139`` VideoMAEEmbeddings.forward        -> embed
* ``src/models/videomae/modeling_videomae.py:179-192`` VideoMAEPatchEmbeddings.forward   -> patch_embed
* ``src/models/videomae/modeling_videomae.py:196-223`` eager_attention_forward           -> attention
* ``src/models/videomae/modeling_videomae.py:258-296`` VideoMAESelfAttention.forward     -> attention
* ``src/models/videomae/modeling_videomae.py:405-431`` VideoMAELayer.forward             -> layer
* ``src/models/videomae/modeling_videomae.py:684-726`` VideoMAEDecoder.forward           -> decoder
* ``src/models/videomae/modeling_videomae.py:753-908`` VideoMAEForPreTraining.forward    -> mim_forward
* ``src/models/videomae/modeling_videomae.py:495-505`` _init_weights                     -> init_params

Pinning: the reference's own tests hold NO value-level fixture for this path; the
oracle is pinned against outputs of the reference's model class (``transformers.VideoMAEForPreTraining``
as called by ``src/run_mim.py:13-21,345-357``) executed in the build container, committed as
``tests/golden/mim_tiny_p16.safetensors`` / ``mim_small_p8.safetensors`` by ``tests/golden/make_golden.py``
(checked by ``tests/test_oracle_cpu.py``: loss 1e-5, logits 1e-4, every gradient 1e-4 relative).

Parameter names are the pinned-era (vendored) state-dict names.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class MIMConfig:
    """Mirror of the ``VideoMAEConfig`` fields the path reads."""

    image_size: int = 512
    num_frames: int = 320
    patch_size: int = 16
    tubelet_size: int = 16
    num_channels: int = 1
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    decoder_hidden_size: int = 384
    decoder_num_hidden_layers: int = 4
    decoder_num_attention_heads: int = 6
    decoder_intermediate_size: int = 1536
    layer_norm_eps: float = 1e-12
    norm_pix_loss: bool = True
    initializer_range: float = 0.02

    @property
    def num_patches(self) -> int:
        g = self.image_size // self.patch_size
        return g * g * (self.num_frames // self.tubelet_size)

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.tubelet_size * self.patch_size ** 2

    def to_dict(self):
        return asdict(self)


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """float64 numpy, sin on even / cos on odd columns, cast to fp32 (spec :95-106), vectorised."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    denom = np.power(10000, 2 * (j // 2) / d_hid)
    tab = pos / denom[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.from_numpy(tab.astype(np.float32))


def layer_param_names(prefix: str):
    a = prefix + "attention.attention."
    return [
        prefix + "layernorm_before.weight", prefix + "layernorm_before.bias",
        a + "query.weight", a + "key.weight", a + "value.weight", a + "q_bias", a + "v_bias",
        prefix + "attention.output.dense.weight", prefix + "attention.output.dense.bias",
        prefix + "layernorm_after.weight", prefix + "layernorm_after.bias",
        prefix + "intermediate.dense.weight", prefix + "intermediate.dense.bias",
        prefix + "output.dense.weight", prefix + "output.dense.bias",
    ]


def param_shapes(cfg: MIMConfig) -> "dict[str, tuple]":
    """Pinned-era state-dict names -> shapes, in canonical (forward) order."""
    d, ff, dd, dff = cfg.hidden_size, cfg.intermediate_size, cfg.decoder_hidden_size, cfg.decoder_intermediate_size
    ts, ps, c = cfg.tubelet_size, cfg.patch_size, cfg.num_channels
    out: "dict[str, tuple]" = {}
    out["videomae.embeddings.patch_embeddings.projection.weight"] = (d, c, ts, ps, ps)
    out["videomae.embeddings.patch_embeddings.projection.bias"] = (d,)

    def layer(prefix, w, f):
        shapes = [(w,), (w,), (w, w), (w, w), (w, w), (w,), (w,), (w, w), (w,), (w,), (w,), (f, w), (f,), (w, f), (w,)]
        for n, s in zip(layer_param_names(prefix), shapes):
            out[n] = s

    for i in range(cfg.num_hidden_layers):
        layer(f"videomae.encoder.layer.{i}.", d, ff)
    out["encoder_to_decoder.weight"] = (dd, d)
    out["mask_token"] = (1, 1, dd)
    for i in range(cfg.decoder_num_hidden_layers):
        layer(f"decoder.decoder_layers.{i}.", dd, dff)
    out["decoder.norm.weight"] = (dd,)
    out["decoder.norm.bias"] = (dd,)
    out["decoder.head.weight"] = (cfg.patch_dim, dd)
    out["decoder.head.bias"] = (cfg.patch_dim,)
    return out


def init_params(cfg: MIMConfig, seed: int = 42, nonzero_small: bool = False) -> "dict[str, torch.Tensor]":
    """_init_weights semantics (spec :495-505): Linear/Conv3d normal(0, initializer_range), biases 0,
    LayerNorm 1/0, mask_token and q_bias/v_bias 0.  ``nonzero_small`` perturbs the zero/one tensors so that
    parity tests exercise every bias / gain path."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("weight") and len(shape) >= 2:
            t = torch.randn(shape, generator=g) * cfg.initializer_range
        elif "layernorm" in name or name.startswith("decoder.norm"):
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            if nonzero_small:
                t = t + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape)
            if nonzero_small:
                t = t + 0.02 * torch.randn(shape, generator=g)
        p[name] = t
    return p


def _r(x: torch.Tensor, emulate_bf16: bool) -> torch.Tensor:
    """Round to bf16 and back (differentiable straight-through as autocast does: grads flow in fp32 here)."""
    if not emulate_bf16:
        return x
    return x.to(torch.bfloat16).to(torch.float32)


def _linear(x, w, b, emu):
    return F.linear(_r(x, emu), _r(w, emu), b)


def patchify(pixel_values: torch.Tensor, cfg: MIMConfig) -> torch.Tensor:
    """[B,T,C,H,W] -> [B,N,ts*ps*ps*C] with per-patch element order (ts,ps,ps,C) (spec :835-867) and token
    order (T',H',W') row-major."""
    B, T, C, H, W = pixel_values.shape
    ts, ps = cfg.tubelet_size, cfg.patch_size
    f = pixel_values.view(B, T // ts, ts, C, H // ps, ps, W // ps, ps)
    f = f.permute(0, 1, 4, 6, 2, 5, 7, 3).contiguous()
    return f.view(B, (T // ts) * (H // ps) * (W // ps), ts * ps * ps, C)


def labels_from_pixels(pixel_values: torch.Tensor, cfg: MIMConfig) -> torch.Tensor:
    """Per-patch normalised targets, all N patches (spec :835-867): (x-mean)/(sqrt(var_unbiased)+1e-6)."""
    frames = patchify(pixel_values, cfg)
    if cfg.norm_pix_loss:
        frames = (frames - frames.mean(dim=-2, keepdim=True)) / (frames.var(dim=-2, unbiased=True, keepdim=True).sqrt() + 1e-6)
    elif cfg.num_channels != 3:
        raise ValueError("Can't unnormalize non-RGB images. Consider setting config.norm_pix_loss to False.")
    B, N = frames.shape[:2]
    return frames.view(B, N, -1)


def patch_embed(pixel_values, w, b, cfg: MIMConfig, emu=False):
    """Conv3d with kernel==stride as a matmul over bricks (spec :172-177,190-191). Token order (T',H',W')."""
    B, T, C, H, W = pixel_values.shape
    if C != cfg.num_channels:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
    if H != cfg.image_size or W != cfg.image_size:
        raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")
    ts, ps = cfg.tubelet_size, cfg.patch_size
    # conv weight [d, C, ts, ps, ps] flattens to k = ((c*ts + kt)*ps + kh)*ps + kw
    x = pixel_values.view(B, T // ts, ts, C, H // ps, ps, W // ps, ps)
    x = x.permute(0, 1, 4, 6, 3, 2, 5, 7).contiguous().view(B, -1, C * ts * ps * ps)
    return _linear(x, w.reshape(w.shape[0], -1), b, emu)


SDPA = False  # bench.py's cpu_baseline leg sets this: same function through torch's fused CPU kernel (the reference's default
              # `_attn_implementation` is "sdpa", dispatched at spec :270-289); the explicit form below is the parity oracle.


def attention(x, p, prefix, heads, emu=False):
    """q/k/v linears (k bias == 0), fp32 softmax(QK^T * Dh^-0.5) V (spec :196-223, :258-296)."""
    a = prefix + "attention.attention."
    B, S, D = x.shape
    dh = D // heads
    q = _linear(x, p[a + "query.weight"], p[a + "q_bias"], emu)
    k = _linear(x, p[a + "key.weight"], None, emu)
    v = _linear(x, p[a + "value.weight"], p[a + "v_bias"], emu)
    q, k, v = (_r(t, emu).view(B, S, heads, dh).transpose(1, 2) for t in (q, k, v))
    if SDPA and not emu:
        return F.scaled_dot_product_attention(q, k, v, scale=dh ** -0.5).transpose(1, 2).reshape(B, S, D)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    pr = torch.softmax(s, dim=-1, dtype=torch.float32)
    o = torch.matmul(_r(pr, emu), v).transpose(1, 2).reshape(B, S, D)
    return _r(o, emu)


def layer(x, p, prefix, heads, eps, emu=False):
    """Pre-LN block (spec :405-431)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_before.weight"], p[prefix + "layernorm_before.bias"], eps)
    a = attention(h, p, prefix, heads, emu)
    a = _linear(a, p[prefix + "attention.output.dense.weight"], p[prefix + "attention.output.dense.bias"], emu)
    x = a + x
    h = F.layer_norm(x, (D,), p[prefix + "layernorm_after.weight"], p[prefix + "layernorm_after.bias"], eps)
    h = _linear(h, p[prefix + "intermediate.dense.weight"], p[prefix + "intermediate.dense.bias"], emu)
    h = F.gelu(_r(h, emu))  # exact erf GELU (ACT2FN["gelu"])
    h = _linear(h, p[prefix + "output.dense.weight"], p[prefix + "output.dense.bias"], emu)
    return h + x


def mim_forward(p: "dict[str, torch.Tensor]", cfg: MIMConfig, pixel_values: torch.Tensor,
                bool_masked_pos: torch.Tensor, emulate_bf16: bool = False, return_intermediates: bool = False):
    """VideoMAEForPreTraining.forward (spec :753-908).  Returns (loss, logits[, dict])."""
    emu = emulate_bf16
    if bool_masked_pos is None:
        raise ValueError("One must provided a boolean mask ")
    B = pixel_values.shape[0]
    N, d, dd = cfg.num_patches, cfg.hidden_size, cfg.decoder_hidden_size
    inter = {}
    emb = patch_embed(pixel_values, p["videomae.embeddings.patch_embeddings.projection.weight"],
                      p["videomae.embeddings.patch_embeddings.projection.bias"], cfg, emu)
    emb = emb + sinusoid_table(N, d)[None]
    x = emb[~bool_masked_pos].reshape(B, -1, d)  # visible tokens, ascending index order
    inter["embeddings_visible"] = x
    for i in range(cfg.num_hidden_layers):
        x = layer(x, p, f"videomae.encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, emu)
        if i == 0:
            inter["encoder_layer0_out"] = x
    inter["encoder_out"] = x
    x = _linear(x, p["encoder_to_decoder.weight"], None, emu)
    pos = sinusoid_table(N, dd)[None].expand(B, -1, -1)
    pos_vis = pos[~bool_masked_pos].reshape(B, -1, dd)
    pos_mask = pos[bool_masked_pos].reshape(B, -1, dd)
    x_full = torch.cat([x + pos_vis, p["mask_token"] + pos_mask], dim=1)  # visible first, masked last
    inter["decoder_in"] = x_full
    n_mask = pos_mask.shape[1]
    for i in range(cfg.decoder_num_hidden_layers):
        x_full = layer(x_full, p, f"decoder.decoder_layers.{i}.", cfg.decoder_num_attention_heads, cfg.layer_norm_eps, emu)
    h = x_full[:, -n_mask:]
    h = F.layer_norm(h, (dd,), p["decoder.norm.weight"], p["decoder.norm.bias"], 1e-5)  # torch default eps (spec :676)
    logits = _linear(h, p["decoder.head.weight"], p["decoder.head.bias"], emu)
    logits = _r(logits, emu)
    with torch.no_grad():
        labels_all = labels_from_pixels(pixel_values, cfg)
        labels = labels_all[bool_masked_pos].reshape(B, -1, labels_all.shape[-1])
    loss = F.mse_loss(logits, labels)
    if return_intermediates:
        inter["labels"] = labels
        return loss, logits, inter
    return loss, logits


def encoder_forward(p, cfg: MIMConfig, pixel_values, bool_masked_pos=None, prefix="videomae.", final_ln=False, emulate_bf16=False, hidden_states=None):
    """VideoMAEModel.forward (spec :509-655; embeddings :124-139): all tokens when bool_masked_pos is None (the embedding export of
    src/run_inference.py:78-83), the visible ones otherwise; final ``layernorm`` only when the config has use_mean_pooling=False (:517-520).
    ``hidden_states`` (a list) receives what output_hidden_states=True collects (spec :452-457, 478-483): the embedding output and every layer's output."""
    emu = emulate_bf16
    B = pixel_values.shape[0]
    N, d = cfg.num_patches, cfg.hidden_size
    emb = patch_embed(pixel_values, p[prefix + "embeddings.patch_embeddings.projection.weight"],
                      p[prefix + "embeddings.patch_embeddings.projection.bias"], cfg, emu)
    x = emb + sinusoid_table(N, d)[None]
    if bool_masked_pos is not None:
        x = x[~bool_masked_pos].reshape(B, -1, d)
    if hidden_states is not None:
        hidden_states.append(x)
    for i in range(cfg.num_hidden_layers):
        x = layer(x, p, f"{prefix}encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, emu)
        if hidden_states is not None:
            hidden_states.append(x)
    if final_ln:
        x = F.layer_norm(x, (d,), p[prefix + "layernorm.weight"], p[prefix + "layernorm.bias"], cfg.layer_norm_eps)
    return x


def classification_forward(p, cfg: MIMConfig, pixel_values, additional_features=None, emulate_bf16=False):
    """VideoMAEForVideoClassification.forward with use_mean_pooling=True (spec :917-989): mean over tokens -> fc_norm (torch default
    eps 1e-5, :925) -> [concat additional_features] -> classifier."""
    seq = encoder_forward(p, cfg, pixel_values, None, "videomae.", False, emulate_bf16)
    pooled = F.layer_norm(seq.mean(1), (cfg.hidden_size,), p["fc_norm.weight"], p["fc_norm.bias"], 1e-5)
    if additional_features is not None:
        pooled = torch.cat([pooled, additional_features], dim=-1)
    return F.linear(pooled, p["classifier.weight"], p["classifier.bias"])


def mim_step(p, cfg, pixel_values, bool_masked_pos, emulate_bf16=False):
    """Forward + backward; returns (loss, logits, grads dict) — what one train step hands the optimizer."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, logits = mim_forward(leaf, cfg, pixel_values, bool_masked_pos, emulate_bf16)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), logits.detach(), grads


# ----------------------------------------------------------------------------------------------------------
# reference-shaped per-op helpers used by the per-kernel parity tests
# ----------------------------------------------------------------------------------------------------------

def gelu_exact(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def attention_core(q, k, v, scale):
    """softmax_fp32(q k^T * scale) v for [B,H,S,Dh] tensors; returns (out, lse)."""
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    lse = torch.logsumexp(s, dim=-1)
    return torch.matmul(torch.softmax(s, dim=-1), v.float()), lse
