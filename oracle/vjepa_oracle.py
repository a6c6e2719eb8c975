"""CPU oracle for the V-JEPA train step (``src/run_vjepa.py``).  TEST INFRASTRUCTURE ONLY (same rules as mim_oracle.py).

Plain PyTorch fp32 restatement of the reference arithmetic; each function cites the lines it follows
(relative to /root/reference):

* ``src/models/vjepa/modeling_vjepa.py:104-170``  patch embedding (Conv3d, no position table)          -> patch_embed
* ``src/models/vjepa/modeling_vjepa.py:204-228``  rotate_queries_or_keys (tiled cos/sin, interleaved pairs) -> rotate
* ``src/models/vjepa/modeling_vjepa.py:277-326``  position ids -> (frame, height, width), 3 RoPE segments  -> rope3d
* ``src/models/vjepa/modeling_vjepa.py:328-371``  VJEPA2RopeAttention.forward                             -> attention
* ``src/models/vjepa/modeling_vjepa.py:410-478``  VJEPA2MLP / VJEPA2Layer                                  -> layer
* ``src/models/vjepa/modeling_vjepa.py:507-539``  VJEPA2Encoder.forward                                    -> encoder
* ``src/models/vjepa/modeling_vjepa.py:542-556``  apply_masks                                              -> apply_masks
* ``src/models/vjepa/modeling_vjepa.py:588-626``  VJEPA2PredictorEmbeddings.forward                        -> predictor
* ``src/models/vjepa/modeling_vjepa.py:658-746``  sort / unsort / VJEPA2Predictor.forward                  -> predictor
* ``src/models/vjepa/modeling_vjepa.py:1081-1163`` VJEPA2Model.forward                                     -> model_forward
* ``src/run_vjepa.py:87-141``                      MomentumEncoder.update, VJEPATrainer.compute_loss       -> ema_update, vjepa_loss
* ``src/dataloader/transforms.py:96-217``          VJEPAMaskGenerator                                      -> VJEPAMaskGeneratorOracle

Pinned against outputs of the reference classes executed in the build container
(``tests/golden/vjepa_tiny.safetensors``, ``vjepa_masks.npz``; ``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class VJEPAConfig:
    patch_size: int = 16
    crop_size: int = 512
    frames_per_clip: int = 320
    tubelet_size: int = 16
    in_chans: int = 1
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    mlp_ratio: float = 4.0
    layer_norm_eps: float = 1e-6
    pred_hidden_size: int = 384
    pred_num_hidden_layers: int = 12
    pred_num_attention_heads: int = 12
    pred_num_mask_tokens: int = 10
    pred_mlp_ratio: float = 4.0
    initializer_range: float = 0.02

    @property
    def grid_size(self):
        return self.crop_size // self.patch_size

    @property
    def grid_depth(self):
        return self.frames_per_clip // self.tubelet_size

    @property
    def num_patches(self):
        return self.grid_depth * self.grid_size * self.grid_size


def _layer_shapes(prefix, w, ff):
    out = {}
    out[prefix + "norm1.weight"] = (w,)
    out[prefix + "norm1.bias"] = (w,)
    for n in ("query", "key", "value", "proj"):
        out[prefix + f"attention.{n}.weight"] = (w, w)
        out[prefix + f"attention.{n}.bias"] = (w,)
    out[prefix + "norm2.weight"] = (w,)
    out[prefix + "norm2.bias"] = (w,)
    out[prefix + "mlp.fc1.weight"] = (ff, w)
    out[prefix + "mlp.fc1.bias"] = (ff,)
    out[prefix + "mlp.fc2.weight"] = (w, ff)
    out[prefix + "mlp.fc2.bias"] = (w,)
    return out


def param_shapes(cfg: VJEPAConfig):
    """Reference state-dict names -> shapes (encoder + predictor; pooler/classifier are off the path)."""
    d, pd = cfg.hidden_size, cfg.pred_hidden_size
    out = {"encoder.embeddings.patch_embeddings.proj_3d.weight": (d, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size),
           "encoder.embeddings.patch_embeddings.proj_3d.bias": (d,)}
    for i in range(cfg.num_hidden_layers):
        out.update(_layer_shapes(f"encoder.layer.{i}.", d, int(d * cfg.mlp_ratio)))
    out["encoder.layernorm.weight"] = (d,)
    out["encoder.layernorm.bias"] = (d,)
    out["predictor.embeddings.mask_tokens"] = (cfg.pred_num_mask_tokens, 1, 1, pd)
    out["predictor.embeddings.predictor_embeddings.weight"] = (pd, d)
    out["predictor.embeddings.predictor_embeddings.bias"] = (pd,)
    for i in range(cfg.pred_num_hidden_layers):
        out.update(_layer_shapes(f"predictor.layer.{i}.", pd, int(pd * cfg.pred_mlp_ratio)))
    out["predictor.layernorm.weight"] = (pd,)
    out["predictor.layernorm.bias"] = (pd,)
    out["predictor.proj.weight"] = (d, pd)
    out["predictor.proj.bias"] = (d,)
    return out


def init_params(cfg: VJEPAConfig, seed=0, nonzero_small=True):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("weight") and len(shape) >= 2:
            t = torch.randn(shape, generator=g) * cfg.initializer_range
        elif "norm" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            if nonzero_small:
                t = t + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.zeros(shape)
            if nonzero_small:
                t = t + 0.02 * torch.randn(shape, generator=g)
        p[name] = t
    return p


def rotate(x, pos):
    """rotate_queries_or_keys (:204-228): x [B,H,N,D], pos broadcastable to [..., N]; cos/sin TILED ([f0..f_{D/2-1}] twice)."""
    D = x.shape[-1]
    omega = torch.arange(D // 2, dtype=x.dtype)
    omega /= D / 2.0
    omega = 1.0 / 10000 ** omega
    freq = pos.to(x.dtype).unsqueeze(-1) * omega
    s = freq.sin().repeat_interleave(1, -1)
    c = freq.cos()
    s = torch.cat([s, s], dim=-1)
    c = torch.cat([c, c], dim=-1)
    y = x.unflatten(-1, (-1, 2))
    y1, y2 = y.unbind(dim=-1)
    y = torch.stack((-y2, y1), dim=-1).flatten(-2)
    return x * c + y * s


def rope3d(qk, ids, grid_size):
    """apply_rotary_embeddings + get_position_ids (:277-326). qk [B,H,N,Dh]; ids int64 [N] or [B,N]."""
    Dh = qk.shape[-1]
    seg = int(2 * ((Dh // 3) // 2))
    tpf = grid_size * grid_size
    if ids.dim() == 2:
        ids = ids.unsqueeze(1)  # [B,1,N] broadcast over heads
    frame = ids // tpf
    height = (ids - tpf * frame) // grid_size
    width = (ids - tpf * frame) - grid_size * height
    parts = [rotate(qk[..., 0:seg], frame), rotate(qk[..., seg:2 * seg], height), rotate(qk[..., 2 * seg:3 * seg], width)]
    if 3 * seg < Dh:
        parts.append(qk[..., 3 * seg:])
    return torch.cat(parts, dim=-1)


SDPA = False  # full-size checks set this: the same softmax(QK^T/sqrt(Dh))V through torch's fused CPU kernel (no S x S matrix in memory)


def attention(x, p, prefix, heads, ids, grid_size):
    B, S, D = x.shape
    dh = D // heads
    a = prefix + "attention."
    q = F.linear(x, p[a + "query.weight"], p[a + "query.bias"]).view(B, S, heads, dh).transpose(1, 2)
    k = F.linear(x, p[a + "key.weight"], p[a + "key.bias"]).view(B, S, heads, dh).transpose(1, 2)
    v = F.linear(x, p[a + "value.weight"], p[a + "value.bias"]).view(B, S, heads, dh).transpose(1, 2)
    k = rope3d(k, ids, grid_size)
    q = rope3d(q, ids, grid_size)
    if SDPA:
        o = F.scaled_dot_product_attention(q, k, v, scale=dh ** -0.5).transpose(1, 2).reshape(B, S, D)
        return F.linear(o, p[a + "proj.weight"], p[a + "proj.bias"])
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), v).transpose(1, 2).reshape(B, S, D)
    return F.linear(o, p[a + "proj.weight"], p[a + "proj.bias"])


def layer(x, p, prefix, heads, eps, ids, grid_size):
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    x = attention(h, p, prefix, heads, ids, grid_size) + x
    h = F.layer_norm(x, (D,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    h = F.linear(h, p[prefix + "mlp.fc1.weight"], p[prefix + "mlp.fc1.bias"])
    h = F.linear(F.gelu(h), p[prefix + "mlp.fc2.weight"], p[prefix + "mlp.fc2.bias"])
    return h + x


def patch_embed(pix, w, b, cfg: VJEPAConfig):
    """[B,T,C,H,W] -> [B,N,d]; token order (T',H',W') row-major (:134-136, :154-170)."""
    x = F.conv3d(pix.permute(0, 2, 1, 3, 4), w, b, stride=(cfg.tubelet_size, cfg.patch_size, cfg.patch_size))
    return x.flatten(2).transpose(1, 2)


def encoder_body(p, cfg: VJEPAConfig, pix):
    """Embeddings + the L layers, WITHOUT the final LayerNorm (:507-529)."""
    x = patch_embed(pix, p["encoder.embeddings.patch_embeddings.proj_3d.weight"], p["encoder.embeddings.patch_embeddings.proj_3d.bias"], cfg)
    ids = torch.arange(x.shape[1])
    for i in range(cfg.num_hidden_layers):
        x = layer(x, p, f"encoder.layer.{i}.", cfg.num_attention_heads, cfg.layer_norm_eps, ids, cfg.grid_size)
    return x


def encoder(p, cfg: VJEPAConfig, pix):
    x = encoder_body(p, cfg, pix)
    return F.layer_norm(x, (cfg.hidden_size,), p["encoder.layernorm.weight"], p["encoder.layernorm.bias"], cfg.layer_norm_eps)


def apply_masks(t, masks):
    return torch.cat([torch.gather(t, 1, m.unsqueeze(-1).expand(-1, -1, t.size(-1))) for m in masks], dim=0)


def predictor(p, cfg: VJEPAConfig, enc_out, context_mask, target_mask, mask_index=1):
    ctx = apply_masks(enc_out, context_mask)
    B, n_ctx, _ = ctx.shape
    ctx = F.linear(ctx, p["predictor.embeddings.predictor_embeddings.weight"], p["predictor.embeddings.predictor_embeddings.bias"])
    tok = p["predictor.embeddings.mask_tokens"][mask_index % cfg.pred_num_mask_tokens]
    max_patch_num = int(target_mask[0].max()) + 1
    tgt = apply_masks(tok.repeat(B, max_patch_num, 1), target_mask)
    x = torch.cat([ctx, tgt], dim=1)
    pos = torch.cat([torch.cat(context_mask, 0), torch.cat(target_mask, 0)], dim=1)
    order = torch.argsort(pos, dim=1)
    pos = torch.gather(pos, 1, order)
    x = torch.gather(x, 1, order.unsqueeze(-1).expand(-1, -1, x.size(-1)))
    for i in range(cfg.pred_num_hidden_layers):
        x = layer(x, p, f"predictor.layer.{i}.", cfg.pred_num_attention_heads, cfg.layer_norm_eps, pos, cfg.grid_size)
    x = F.layer_norm(x, (cfg.pred_hidden_size,), p["predictor.layernorm.weight"], p["predictor.layernorm.bias"], cfg.layer_norm_eps)
    rev = torch.argsort(order, dim=1)
    x = torch.gather(x, 1, rev.unsqueeze(-1).expand(-1, -1, x.size(-1)))[:, n_ctx:]
    return F.linear(x, p["predictor.proj.weight"], p["predictor.proj.bias"])


def model_forward(p, cfg: VJEPAConfig, pix, context_mask=None, target_mask=None, skip_predictor=False):
    """VJEPA2Model.forward (:1081-1163): returns dict(last_hidden_state, masked_hidden_state, target_hidden_state, predictor_output)."""
    enc = encoder(p, cfg, pix)
    if context_mask is None and target_mask is None:
        B, N = enc.shape[:2]
        context_mask = [torch.arange(N).unsqueeze(0).repeat(B, 1)]
        target_mask = [torch.arange(N).unsqueeze(0).repeat(B, 1)]
    out = {"last_hidden_state": enc, "masked_hidden_state": apply_masks(enc, context_mask), "target_hidden_state": apply_masks(enc, target_mask),
           "predictor_output": None}
    if not skip_predictor:
        out["predictor_output"] = predictor(p, cfg, enc, context_mask, target_mask)
    return out


def vjepa_loss(p_online, p_target, cfg, pix, context_mask, target_mask):
    """VJEPATrainer.compute_loss (run_vjepa.py:110-141) without the EMA side effect: L1(pred, target_encoder(...).target_hidden_state)."""
    pred = model_forward(p_online, cfg, pix, context_mask, target_mask)["predictor_output"]
    with torch.no_grad():
        tgt = model_forward(p_target, cfg, pix, context_mask, target_mask, skip_predictor=True)["target_hidden_state"]
    return F.l1_loss(pred, tgt), pred, tgt


def vjepa_step(p_online, p_target, cfg, pix, context_mask, target_mask):
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p_online.items()}
    loss, pred, tgt = vjepa_loss(leaf, p_target, cfg, pix, context_mask, target_mask)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    return loss.detach(), pred.detach(), tgt, grads


def ema_update(p_target, p_online, momentum=0.99925):
    """MomentumEncoder.update (run_vjepa.py:94-98): k <- m*k + (1-m)*q, parameter by parameter in iteration order."""
    return {k: p_target[k] * momentum + p_online[k] * (1.0 - momentum) for k in p_target}


class VJEPAMaskGeneratorOracle:
    """VJEPAMaskGenerator (transforms.py:96-217), default flags (no max_keep / inv_block / complements): same RNG call order
    (global torch RNG for the seed and the block origins, a local Generator for scale and aspect ratio)."""

    def __init__(self, input_size=(224, 224, 160), patch_size=(16, 16, 16), pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=1):
        self.depth, self.height, self.width = (input_size[i] // patch_size[i] for i in range(3))
        self.pred_mask_scale, self.aspect_ratio, self.num_blocks = pred_mask_scale, aspect_ratio, num_blocks

    def __call__(self):
        seed = torch.randint(0, 2 ** 32, (1,)).item()
        g = torch.Generator()
        g.manual_seed(seed)
        lo, hi = self.pred_mask_scale
        scale = lo + torch.rand(1, generator=g).item() * (hi - lo)
        num_keep = int(self.depth * self.height * self.width * scale)
        lo, hi = self.aspect_ratio
        ar = lo + torch.rand(1, generator=g).item() * (hi - lo)
        d = int(round(math.pow(num_keep * ar * (1.0 / ar), 1 / 3)))
        h = int(round(d * ar))
        w = int(round(d * (1.0 / ar)))
        d, h, w = min(d, self.depth), min(h, self.height), min(w, self.width)
        keep = torch.ones((self.depth, self.height, self.width), dtype=torch.int32)
        for _ in range(self.num_blocks):
            sd = int(torch.randint(0, self.depth - d + 1, (1,)))
            sh = int(torch.randint(0, self.height - h + 1, (1,)))
            sw = int(torch.randint(0, self.width - w + 1, (1,)))
            for a in range(sd, sd + d):
                for b in range(sh, sh + h):
                    for c in range(sw, sw + w):
                        keep[a, b, c] = 0
        flat = keep.flatten()
        context = torch.nonzero(flat).squeeze()
        target = torch.argwhere(flat == 0).squeeze()
        return context, target
