"""``SmbVJEPA2Model`` — MI355X-native drop-in for the model class ``src/run_vjepa.py`` trains, plus the trainer-side pieces.

Host-side mirror of the reference's ``VJEPA2Model`` (``src/models/vjepa/modeling_vjepa.py:1065-1167``; encoder + predictor; the
attentive pooler / classifier of that file are off the pre-training path): ``Cls(config: VJEPA2Config)``, ``forward(
pixel_values_videos, context_mask=[idx], target_mask=[idx], skip_predictor=...)`` -> object with ``last_hidden_state``,
``masked_hidden_state``, ``target_hidden_state`` and ``predictor_output.last_hidden_state``; same state-dict names
(``encoder.embeddings.patch_embeddings.proj_3d.*``, ``encoder.layer.{i}.{norm1,attention.{query,key,value,proj},norm2,mlp.{fc1,fc2}}.*``,
``encoder.layernorm.*``, ``predictor.embeddings.{predictor_embeddings.*,mask_tokens}``, ``predictor.layer.{i}.*``,
``predictor.layernorm.*``, ``predictor.proj.*``) and the same ``parameters()`` order, which ``MomentumEncoder.update`` zips
(``run_vjepa.py:97``).  The arithmetic runs in libsmbv.so (``smbv_vjepa_forward`` / ``smbv_vjepa_backward``); no CPU fallback.

``vjepa_loss`` / ``MomentumEncoder`` mirror ``VJEPATrainer.compute_loss`` and ``MomentumEncoder.update`` (``run_vjepa.py:87-141``)
with the L1 loss and the EMA as HIP kernels (``smbv_l1_*``, ``smbv_ema_update``).
"""
from __future__ import annotations

import copy
import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
from torch import nn
from transformers import PreTrainedModel, VJEPA2Config
from transformers.utils import ModelOutput

from . import _lib
from .modeling_videomae import _residual_wanted
from .optim import CHUNK


@dataclass
class SmbVJEPA2PredictorOutput(ModelOutput):
    last_hidden_state: Optional[torch.Tensor] = None
    target_hidden_state: Optional[torch.Tensor] = None


@dataclass
class SmbVJEPA2ModelOutput(ModelOutput):
    last_hidden_state: Optional[torch.Tensor] = None
    masked_hidden_state: Optional[torch.Tensor] = None
    target_hidden_state: Optional[torch.Tensor] = None
    predictor_output: Optional[SmbVJEPA2PredictorOutput] = None


def index_plumbing(cms, tms, num_patches, needed=None):
    """Sorted predictor sequences (modeling_vjepa.py:718-719).  Plain index arithmetic on whatever device the masks live on: ``run_vjepa.collate_fn``
    runs it on the host, in the DataLoader's workers, and ships the result with the batch (``smbv_plumbing_*``), so that a step does not open with four
    device sorts and ~60 tiny kernels; ``SmbVJEPA2Model.forward`` runs it on the device when no such arrays come with the call.

    ``cms`` / ``tms``: the mask lists; ``apply_masks`` (:542-556) concatenates their elements along the batch, so predictor sequence m * B + b pairs volume b with mask pair m.  Returns the arrays of
    ``smbv_vjepa_io``: absolute encoder rows of the context tokens, sorted positions, and the slot of every context / target token.

    ``needed`` ("context" / "target", one mask pair): the encoder's token order is permuted so that the tokens whose encoder output the caller
    consumes come first (ascending ids), the others after them (``smbv_vjepa_cfg.enc_needed``); RoPE then takes explicit position ids.  The
    rows ``ctx_idx`` names, and ``need_rows`` (where each consumed token sits), are rows of the COMPACT encoder output [B * enc_needed, d]."""
    B, N = cms[0].shape[0], num_patches
    device = cms[0].device
    cm, tm = torch.cat(cms, dim=0), torch.cat(tms, dim=0)           # [M*B, Nctx], [M*B, Ntgt]
    pos = torch.cat([cm, tm], dim=1)
    n_ctx = cm.shape[1]
    # Slots of the predictor sequence.  The reference sorts the tokens by position and unsorts the output (modeling_vjepa.py:718-746); with RoPE on
    # explicit position ids any order is the same function.  Targets first (by position), then the context (by position): the predictor's output is
    # read at the target tokens only, so its last block works on slots 0 .. Ntgt-1 alone (smbv_vjepa_cfg.pred_tgt_first).
    key = pos + torch.cat([torch.full_like(cm, N), torch.zeros_like(tm)], dim=1)
    order = torch.argsort(key, dim=1, stable=True)
    rev = torch.argsort(order, dim=1, stable=True)
    vol = torch.arange(B, device=device).repeat(len(cms)).unsqueeze(1)  # sequence -> volume it reads
    out = dict(tgt_idx=tm.to(torch.int32).contiguous(), pred_pos=torch.gather(pos, 1, order).to(torch.int32).contiguous(),
               ctx_slot=rev[:, :n_ctx].to(torch.int32).contiguous(), tgt_slot=rev[:, n_ctx:].to(torch.int32).contiguous(), n_masks=len(cms),
               ctx_tok=[c.to(torch.int32).contiguous() for c in cms], tgt_tok=[t.to(torch.int32).contiguous() for t in tms],
               enc_needed=0, enc_pos=None, need_rows=None)
    if needed is None:
        out["all_idx"] = torch.arange(N, device=device, dtype=torch.int32).repeat(B, 1).contiguous()
        out["ctx_idx"] = (cm + vol * N).to(torch.int32).contiguous()
        return out
    sel = cm if needed == "context" else tm                         # [B, n] token ids whose encoder output is consumed (one mask pair)
    n = sel.shape[1]
    rest = torch.ones(B, N, dtype=torch.int8, device=device)
    rest.scatter_(1, sel, 0)                                        # 0 = consumed, 1 = not; a stable sort puts the consumed ids first, both ascending
    perm = torch.argsort(rest, dim=1, stable=True)                  # encoder row j of a volume holds token perm[j]
    rank = torch.argsort(perm, dim=1)                               # token id -> its encoder row (rows of consumed tokens are < n)
    row0 = torch.arange(B, device=device).unsqueeze(1) * n
    out["all_idx"] = perm.to(torch.int32).contiguous()
    out["enc_pos"] = out["all_idx"]
    out["enc_needed"] = n
    out["need_rows"] = torch.gather(rank, 1, sel).to(torch.int32).contiguous()       # [B, n] row inside the volume's compact block
    out["ctx_idx"] = (torch.gather(rank, 1, cm) + row0).to(torch.int32).contiguous() if needed == "context" else (cm + vol * N).to(torch.int32).contiguous()
    return out


def _plumbing_fits(pl, needed, cms, tms):
    """A precomputed index_plumbing() result is used only if it was built for this call's restriction and mask shapes."""
    want = 0 if needed is None else cms[0].shape[1] if needed == "context" else tms[0].shape[1]
    return (int(pl.get("enc_needed", -1)) == want and int(pl.get("n_masks", -1)) == len(cms) and pl["ctx_slot"].shape == (len(cms) * cms[0].shape[0], cms[0].shape[1])
            and pl["tgt_slot"].shape == (len(tms) * tms[0].shape[0], tms[0].shape[1]))


_BLOCK = (("norm1.weight", "ln1_w", "w"), ("norm1.bias", "ln1_b", "w"),
          ("attention.query.weight", "q_w", "ww"), ("attention.query.bias", "q_b", "w"),
          ("attention.key.weight", "k_w", "ww"), ("attention.key.bias", "k_b", "w"),
          ("attention.value.weight", "v_w", "ww"), ("attention.value.bias", "v_b", "w"),
          ("attention.proj.weight", "proj_w", "ww"), ("attention.proj.bias", "proj_b", "w"),
          ("norm2.weight", "ln2_w", "w"), ("norm2.bias", "ln2_b", "w"),
          ("mlp.fc1.weight", "fc1_w", "fw"), ("mlp.fc1.bias", "fc1_b", "f"), ("mlp.fc2.weight", "fc2_w", "wf"), ("mlp.fc2.bias", "fc2_b", "w"))


def param_table(cfg) -> "list[tuple[str, tuple, tuple]]":
    """[(state-dict name, shape, C-struct path)] in the reference's parameters() order."""
    d, pd = cfg.hidden_size, cfg.pred_hidden_size
    ff, pff = int(d * cfg.mlp_ratio), int(pd * cfg.pred_mlp_ratio)
    out = [("encoder.embeddings.patch_embeddings.proj_3d.weight", (d, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size), ("pe_w",)),
           ("encoder.embeddings.patch_embeddings.proj_3d.bias", (d,), ("pe_b",))]

    def block(prefix, grp, i, w, f):
        shp = {"w": (w,), "ww": (w, w), "fw": (f, w), "f": (f,), "wf": (w, f)}
        for suf, field, code in _BLOCK:
            out.append((prefix + suf, shp[code], (grp, i, field)))

    for i in range(cfg.num_hidden_layers):
        block(f"encoder.layer.{i}.", "enc", i, d, ff)
    out += [("encoder.layernorm.weight", (d,), ("enc_ln_w",)), ("encoder.layernorm.bias", (d,), ("enc_ln_b",)),
            ("predictor.embeddings.mask_tokens", (cfg.pred_num_mask_tokens, 1, 1, pd), ("mask_tokens",)),
            ("predictor.embeddings.predictor_embeddings.weight", (pd, d), ("pemb_w",)),
            ("predictor.embeddings.predictor_embeddings.bias", (pd,), ("pemb_b",))]
    for i in range(cfg.pred_num_hidden_layers):
        block(f"predictor.layer.{i}.", "pred", i, pd, pff)
    out += [("predictor.layernorm.weight", (pd,), ("pred_ln_w",)), ("predictor.layernorm.bias", (pd,), ("pred_ln_b",)),
            ("predictor.proj.weight", (d, pd), ("pproj_w",)), ("predictor.proj.bias", (d,), ("pproj_b",))]
    return out


class _Slots(nn.Module):
    """Pure parameter container (reproduces the reference's state-dict names)."""


def _register(root, dotted, p):
    mod = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, _Slots())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], p)


def rope_table(head_dim: int, max_pos: int, device) -> torch.Tensor:
    """fp32 [max_pos][2][seg/2]: cos | sin of pos * 10000^(-j/(seg/2)) (modeling_vjepa.py:204-217 with D = the segment size)."""
    seg = 2 * ((head_dim // 3) // 2)
    half = max(seg // 2, 1)
    omega = 1.0 / 10000 ** (torch.arange(half, dtype=torch.float32) / float(half))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * omega[None, :]
    return torch.stack([ang.cos(), ang.sin()], dim=1).contiguous().to(device)


class _VJEPAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_ref, pix, idx, with_pred, *params):
        model = model_ref()
        enc_out, pred_out, st = model._run_forward(pix, idx, with_pred, grad=True)
        ctx.model_ref, ctx.st = model_ref, st
        ctx.set_materialize_grads(False)
        if pred_out is None:
            pred_out = enc_out.new_zeros(())
            ctx.mark_non_differentiable(pred_out)
        return enc_out, pred_out

    @staticmethod
    def backward(ctx, d_enc, d_pred):
        model = ctx.model_ref()
        n = len(model._table)
        if d_enc is None and d_pred is None:
            return (None,) * (4 + n)
        if d_enc is not None and ctx.st["cfg"].enc_needed:
            raise _lib.SmbvError("smb_vision_amd: gradients through the encoder output need the unrestricted forward (smbv_needed=None)")
        return (None, None, None, None, *model._run_backward(ctx.st, d_pred, d_enc))


class SmbVJEPA2Model(PreTrainedModel):
    config_class = VJEPA2Config
    base_model_prefix = "vjepa2"
    main_input_name = "pixel_values_videos"
    supports_gradient_checkpointing = True
    _supports_sdpa = True
    _supports_flash_attn = True
    _supports_attention_backend = True

    def __init__(self, config: VJEPA2Config):
        super().__init__(config)
        self.config = config
        if config.in_chans != 1:
            raise ValueError("smb_vision_amd: the MI355X V-JEPA path is built for single-channel volumes (in_chans=1, run_vjepa.py:225)")
        if not config.qkv_bias or config.hidden_act != "gelu" or config.drop_path_rate != 0.0:
            raise ValueError("smb_vision_amd: qkv_bias=True, hidden_act='gelu', drop_path_rate=0 only")
        if any(float(getattr(config, k, 0.0) or 0.0) != 0.0 for k in ("attention_probs_dropout_prob", "attention_dropout")):
            raise ValueError("smb_vision_amd: attention dropout must be 0 (not implemented on the HIP path; the reference's configuration leaves it at 0)")
        self._table = param_table(config)
        for name, shape, _ in self._table:
            _register(self, name, nn.Parameter(torch.empty(shape)))
        self.gradient_checkpointing = False
        self.mask_index = 1  # VJEPA2PredictorEmbeddings.forward default (modeling_vjepa.py:593)
        self._gflat = self._structs = self._ws = self._ws_key = self._rope = None
        self._plumbing_checks = []
        self.dp = None  # optional smb_vision_amd.dp.DataParallelEngine (gradient averaging after the backward)
        self._fwd_count = 0
        self.reset_parameters()
        self.post_init()

    # -- parameters ----------------------------------------------------------------------------------------------------
    def reset_parameters(self):
        """_init_weights of the reference (:1015-1046): trunc_normal(std) for Linear/Conv3d weights, zero biases, LayerNorm 1/0,
        mask tokens zero (pred_zero_init_mask_tokens) or trunc_normal."""
        std = self.config.initializer_range
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("mask_tokens"):
                    p.zero_() if self.config.pred_zero_init_mask_tokens else nn.init.trunc_normal_(p, mean=0.0, std=std)
                elif name.endswith("weight") and p.dim() >= 2:
                    nn.init.trunc_normal_(p, mean=0.0, std=std)
                elif "norm" in name and name.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()

    def _init_weights(self, module):
        if not isinstance(module, _Slots) and module is not self:
            return
        names = {id(p): n for n, p in self.named_parameters()}
        std = self.config.initializer_range
        with torch.no_grad():
            for _, p in module.named_parameters(recurse=False):
                if getattr(p, "_is_hf_initialized", False):
                    continue
                n = names.get(id(p), "")
                if n.endswith("mask_tokens"):
                    p.zero_() if self.config.pred_zero_init_mask_tokens else nn.init.trunc_normal_(p, mean=0.0, std=std)
                elif n.endswith("weight") and p.dim() >= 2:
                    nn.init.trunc_normal_(p, mean=0.0, std=std)
                elif "norm" in n and n.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        self.gradient_checkpointing = True  # accepted; activations stay resident

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    def _ordered_params(self):
        sd = dict(self.named_parameters())
        return [sd[n] for n, _, _ in self._table]

    # -- geometry ------------------------------------------------------------------------------------------------------
    @property
    def num_patches(self):
        c = self.config
        return (c.frames_per_clip // c.tubelet_size) * (c.crop_size // c.patch_size) ** 2

    def _cfg_struct(self, B, n_ctx, n_tgt):
        c = self.config
        s = _lib.VjepaCfg()
        s.B, s.T, s.H, s.W, s.ps, s.ts = B, c.frames_per_clip, c.crop_size, c.crop_size, c.patch_size, c.tubelet_size
        s.N, s.Nctx, s.Ntgt = self.num_patches, n_ctx, n_tgt
        s.d, s.heads, s.ff, s.layers = c.hidden_size, c.num_attention_heads, int(c.hidden_size * c.mlp_ratio), c.num_hidden_layers
        s.pd, s.pheads, s.pff, s.players = (c.pred_hidden_size, c.pred_num_attention_heads, int(c.pred_hidden_size * c.pred_mlp_ratio),
                                            c.pred_num_hidden_layers)
        s.ln_eps = c.layer_norm_eps
        return s

    def _ensure_device_state(self, device):
        params = self._ordered_params()
        for p in params:
            if p.device != device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SmbvError("smb_vision_amd: parameters must be contiguous fp32 tensors on the input's HIP device")
        c = self.config
        if self._rope is None or self._rope[0].device != device:
            maxpos = max(c.frames_per_clip // c.tubelet_size, c.crop_size // c.patch_size) + 1
            self._rope = (rope_table(c.hidden_size // c.num_attention_heads, maxpos, device),
                          rope_table(c.pred_hidden_size // c.pred_num_attention_heads, maxpos, device))
        if self._gflat is None or self._gflat.device != device:
            offs, off = [], 0
            for p in params:
                offs.append(off)
                off += (p.numel() + 63) // 64 * 64
            self._goffs = offs
            self._gflat = torch.zeros(off, dtype=torch.float32, device=device)
            self._structs = None
        sig = tuple(p.data_ptr() for p in params)
        if self._structs is None or self._structs[0] != sig:
            ps, gs = _lib.VjepaTensors(), _lib.VjepaTensors()
            gbase = self._gflat.data_ptr()
            pd = c.pred_hidden_size
            mi = self.mask_index % c.pred_num_mask_tokens
            for p, off, (_, _, path) in zip(params, self._goffs, self._table):
                pp, gp = p.data_ptr(), gbase + 4 * off
                if path == ("mask_tokens",):   # only the selected token row takes part; the other rows get zero gradient
                    ps.mask_token, gs.mask_token = pp + 4 * mi * pd, gp + 4 * mi * pd
                elif len(path) == 1:
                    setattr(ps, path[0], pp), setattr(gs, path[0], gp)
                else:
                    setattr(getattr(ps, path[0])[path[1]], path[2], pp), setattr(getattr(gs, path[0])[path[1]], path[2], gp)
            self._structs = (sig, ps, gs)
        return params

    def _queue_plumbing_check(self, idx, cms, tms):
        bad = torch.zeros((), dtype=torch.int32, device=cms[0].device)
        for got, want in zip(list(idx["ctx_tok"]) + list(idx["tgt_tok"]), list(cms) + list(tms)):
            bad = bad + (got.to(want.dtype) != want).any().to(torch.int32)
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(bad.reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._plumbing_checks.append((ev, host, self._fwd_count + 1))

    def _check_pending_plumbing(self, wait=False):
        keep = []
        for ev, host, step in self._plumbing_checks:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, step))
            elif int(host[0]) != 0:
                self._plumbing_checks = []
                raise ValueError(f"smbv_plumbing of forward #{step} was not built from the context / target masks passed with it (index_plumbing() must "
                                 "see the masks the forward receives); its outputs and gradients used the plumbing's masks")
        self._plumbing_checks = keep

    def check_plumbing(self):
        """Blocks until every queued smbv_plumbing / mask comparison has been read back; raises if one failed."""
        self._check_pending_plumbing(wait=True)

    def grad_views(self):
        return [self._gflat[o:o + p.numel()].view(p.shape) for p, o in zip(self._ordered_params(), self._goffs)]

    # -- the step ------------------------------------------------------------------------------------------------------
    def _index_plumbing(self, cms, tms, device, needed=None):
        return index_plumbing(cms, tms, self.num_patches, needed)

    def _run_forward(self, pix, idx, with_pred, grad=False):
        lib = _lib.load()
        dev = pix.device
        B = pix.shape[0]
        n_ctx, n_tgt = idx["ctx_idx"].shape[1], idx["tgt_idx"].shape[1]
        cfg = self._cfg_struct(B, n_ctx, n_tgt)
        cfg.n_masks = idx["n_masks"]
        cfg.enc_needed = idx["enc_needed"]
        cfg.pred_tgt_first = 0 if os.environ.get("SMBV_VJEPA_PRED_FULL", "0") == "1" else 1   # _index_plumbing puts the target slots first (A/B switch)
        # delta = rowsum(dO * O) from the attention output to 16 significant bits (smbv_vjepa_cfg.attn_o_resid): only a forward that will be differentiated
        # pays for the second output (the frozen target encoder never does); config.smbv_attention_residual = False gives the bf16-O delta of flash attention
        cfg.attn_o_resid = int(bool(grad and _residual_wanted(self.config, True)))
        need = lib.smbv_vjepa_workspace_bytes(C.byref(cfg))
        if need == 0:
            raise ValueError("smbv_vjepa_workspace_bytes: " + _lib.last_error())
        key = (B, n_ctx, n_tgt, idx["n_masks"], idx["enc_needed"], dev, cfg.attn_o_resid)
        if self._ws is None or self._ws_key != key or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws_key = key
        d = self.config.hidden_size
        enc_out = torch.empty(B, idx["enc_needed"] or self.num_patches, d, dtype=torch.float32, device=dev)
        pred_out = torch.empty(B * idx["n_masks"], n_tgt, d, dtype=torch.float32, device=dev) if with_pred else None
        io = _lib.VjepaIO()
        io.pixel_values = pix.data_ptr()
        for k in ("all_idx", "ctx_idx", "tgt_idx", "pred_pos", "ctx_slot", "tgt_slot"):
            setattr(io, k, idx[k].data_ptr())
        io.rope_enc, io.rope_pred = self._rope[0].data_ptr(), self._rope[1].data_ptr()
        io.enc_out, io.pred_out, io.d_pred = enc_out.data_ptr(), (pred_out.data_ptr() if with_pred else None), None
        io.enc_pos = idx["enc_pos"].data_ptr() if idx["enc_pos"] is not None else None
        _, ps, gs = self._structs
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.smbv_vjepa_forward(C.byref(cfg), C.byref(ps), C.byref(io), self._ws.data_ptr(), self._ws.numel(), stream), "smbv_vjepa_forward")
        self._fwd_count += 1
        return enc_out, pred_out, dict(cfg=cfg, io=io, step=self._fwd_count, keep=(pix, idx, enc_out, pred_out))

    def _run_backward(self, st, d_pred, d_enc=None):
        lib = _lib.load()
        if st["step"] != self._fwd_count:
            raise _lib.SmbvError("smb_vision_amd: backward() after a later forward() reused the step workspace")
        io = st["io"]
        if d_pred is not None:
            d_pred = d_pred.detach().to(torch.float32).contiguous()
        if d_enc is not None:   # gradient through last_hidden_state / masked_hidden_state / target_hidden_state (modeling_vjepa.py:1137-1163)
            d_enc = d_enc.detach().to(torch.float32).contiguous()
        io.d_pred = d_pred.data_ptr() if d_pred is not None else None
        io.d_enc = d_enc.data_ptr() if d_enc is not None else None
        _, ps, gs = self._structs
        _lib.detach_aliased_grads(self._ordered_params(), self._gflat, self._goffs, self)  # accumulation / zero_grad(set_to_none=False)
        self._gflat.zero_()  # rows of mask_tokens that were not selected keep a zero gradient
        stream = torch.cuda.current_stream().cuda_stream
        events, n_ev = None, 0
        if self.dp is not None:
            self.dp.prepare(self)
            events, n_ev = self.dp.event_array()
        _lib.check(lib.smbv_vjepa_backward(C.byref(st["cfg"]), C.byref(ps), C.byref(gs), C.byref(io), self._ws.data_ptr(), self._ws.numel(), events, n_ev,
                                           stream), "smbv_vjepa_backward")
        if self.dp is not None:  # data parallel: each bucket's all-reduce starts when its last gradient group is final
            self.dp.allreduce_after_backward(self)
        return self.grad_views()

    def get_vision_features(self, pixel_values_videos) -> torch.Tensor:
        """``VJEPA2Model.get_vision_features`` (modeling_vjepa.py:1165-1167): the encoder's ``last_hidden_state`` over every token.  The reference runs its
        predictor as well and throws the result away; the encoder output does not depend on it, so it is skipped here."""
        return self.forward(pixel_values_videos, skip_predictor=True).last_hidden_state

    def get_input_embeddings(self):
        """``VJEPA2Model.get_input_embeddings`` (:1076-1077): the patch-embedding node (here the container of ``proj_3d.{weight,bias}``)."""
        return self._modules["encoder"]._modules["embeddings"]._modules["patch_embeddings"]

    # -- data parallel hooks (smb_vision_amd.dp.DataParallelEngine) --------------------------------------------------------------
    def flat_grads(self) -> torch.Tensor:
        return self._gflat

    def grad_group_ranges(self):
        """Element ranges of the flat gradient arena in the order smbv_vjepa_backward completes them (its event order): predictor.proj +
        predictor.layernorm, predictor layers (last first), mask tokens + predictor_embeddings, encoder.layernorm, encoder layers (last
        first), patch embedding."""
        names = [n for n, _, _ in self._table]
        sizes = [(p.numel() + 63) // 64 * 64 for p in self._ordered_params()]
        offs = self._goffs

        def span(pred):
            idx = [i for i, n in enumerate(names) if pred(n)]
            return (offs[idx[0]], offs[idx[-1]] + sizes[idx[-1]])

        c = self.config
        out = [span(lambda n: n.startswith("predictor.layernorm.") or n.startswith("predictor.proj."))]
        for l in reversed(range(c.pred_num_hidden_layers)):
            out.append(span(lambda n, l=l: n.startswith(f"predictor.layer.{l}.")))
        out.append(span(lambda n: n.startswith("predictor.embeddings.")))
        out.append(span(lambda n: n.startswith("encoder.layernorm.")))
        for l in reversed(range(c.num_hidden_layers)):
            out.append(span(lambda n, l=l: n.startswith(f"encoder.layer.{l}.")))
        out.append(span(lambda n: n.startswith("encoder.embeddings.")))
        return out

    def forward(self, pixel_values_videos: torch.Tensor, context_head_mask=None, context_mask: Optional[List[torch.Tensor]] = None,
                target_head_mask=None, target_mask: Optional[List[torch.Tensor]] = None, skip_predictor: bool = False, output_attentions=None,
                output_hidden_states=None, smbv_needed: Optional[str] = None, smbv_plumbing: Optional[dict] = None, **kwargs):
        """``smbv_needed`` (not in the reference's signature; default None = the reference's forward): "context" - the caller consumes
        ``predictor_output`` / ``masked_hidden_state`` only (the online model inside ``VJEPATrainer.compute_loss``, run_vjepa.py:118-124), "target" - it
        consumes ``target_hidden_state`` only (the frozen target encoder, :126-131).  The last encoder block then computes only the rows those outputs
        are made of (``smbv_vjepa_cfg.enc_needed``); ``last_hidden_state`` and the other gather are ``None``.  What IS returned equals the
        unrestricted forward's output up to summation order (the encoder's tokens are processed in a permuted order: same products, sums in another
        order), tests/test_vjepa_gpu.py."""
        if pixel_values_videos is None:
            raise ValueError("You have to specify pixel_values_videos")
        if output_attentions or output_hidden_states or context_head_mask is not None or target_head_mask is not None:
            raise NotImplementedError("smb_vision_amd: head masks / attention maps / hidden states are not produced by the fused HIP path")
        if not pixel_values_videos.is_cuda:
            raise _lib.SmbvError("smb_vision_amd: pixel_values_videos must live on a HIP device; there is no CPU fallback")
        c = self.config
        B, T, Cc, H, W = pixel_values_videos.shape
        if Cc != c.in_chans or H != c.crop_size or W != c.crop_size or T != c.frames_per_clip:
            raise ValueError(f"pixel_values_videos {tuple(pixel_values_videos.shape)} does not match the config "
                             f"([B,{c.frames_per_clip},{c.in_chans},{c.crop_size},{c.crop_size}])")
        dev = pixel_values_videos.device
        _lib.load()
        with torch.cuda.device(dev):
            pix = pixel_values_videos.to(torch.float32).contiguous()
            N = self.num_patches
            if context_mask is None and target_mask is None:
                ar = torch.arange(N, device=dev).unsqueeze(0).repeat(B, 1)
                context_mask, target_mask = [ar], [ar]
            if len(context_mask) != len(target_mask) or len(context_mask) < 1:
                raise ValueError("context_mask and target_mask must be lists of the same, non-zero length")
            cms, tms = [m.to(dev).long() for m in context_mask], [m.to(dev).long() for m in target_mask]
            if any(m.dim() != 2 or m.shape[0] != B for m in cms + tms):
                raise ValueError("every mask must be an index tensor [batch, tokens]")
            if any(m.shape[1] != cms[0].shape[1] for m in cms) or any(m.shape[1] != tms[0].shape[1] for m in tms):
                raise ValueError("all context masks (and all target masks) must keep the same number of tokens: apply_masks concatenates them "
                                 "along the batch (modeling_vjepa.py:542-556)")
            params = self._ensure_device_state(dev)
            if smbv_needed not in (None, "context", "target"):
                raise ValueError("smbv_needed must be None, 'context' or 'target'")
            sel = {"context": cms, "target": tms}.get(smbv_needed)
            native_heads = (c.hidden_size // c.num_attention_heads) in (32, 64)  # (head-padded encoder widths: the ranged block is not built for them)
            if smbv_needed is not None and (len(cms) != 1 or sel[0].shape[1] >= N or not native_heads or (smbv_needed == "target" and not skip_predictor)):
                smbv_needed = None  # mask lists, a mask that keeps every token, or a predictor that needs the context rows: the unrestricted forward
            self._check_pending_plumbing()
            if smbv_plumbing is not None and _plumbing_fits(smbv_plumbing, smbv_needed, cms, tms):
                # built by index_plumbing() ahead of the step (run_vjepa.collate_fn, on the host): only the uploads remain
                idx = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else ([t.to(dev, non_blocking=True) for t in v] if isinstance(v, (list, tuple)) else v))
                       for k, v in smbv_plumbing.items()}
                # the arrays must have been built from THESE masks (a caller that perturbs the masks after collate, or reuses a cached batch
                # dict, would otherwise run on stale indices silently).  Compared on the device, read back without a synchronisation: a mismatch raises
                # at the next forward (or at check_plumbing()), naming this step.
                self._queue_plumbing_check(idx, cms, tms)
            else:
                idx = self._index_plumbing(cms, tms, dev, smbv_needed)
            with_pred = not skip_predictor
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                enc_out, pred_out = _VJEPAFn.apply(weakref.ref(self), pix, idx, with_pred, *params)
                if not with_pred:
                    pred_out = None
            else:
                enc_out, pred_out, _ = self._run_forward(pix, idx, with_pred)
            # apply_masks(sequence_output, masks) (:1150-1158): one HBM-rate row gather per mask, concatenated along the batch
            if smbv_needed is None:   # (differentiable like the reference's: _GatherRowsFn scatters the gradient back into the encoder output's)
                masked = torch.cat([_GatherRowsFn.apply(enc_out, c) for c in idx["ctx_tok"]], dim=0)
                target = torch.cat([_GatherRowsFn.apply(enc_out, t) for t in idx["tgt_tok"]], dim=0)
                last = enc_out
            else:  # the encoder output is compact [B, enc_needed, d]: the consumed tokens, in ascending id order; need_rows = where each mask entry sits
                got = gather_rows(enc_out.detach(), idx["need_rows"])
                masked, target, last = (got, None, None) if smbv_needed == "context" else (None, got, None)
        pred = SmbVJEPA2PredictorOutput(last_hidden_state=pred_out, target_hidden_state=target) if with_pred else None
        return SmbVJEPA2ModelOutput(last_hidden_state=last, masked_hidden_state=masked, target_hidden_state=target, predictor_output=pred)


def gather_rows(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """apply_masks (modeling_vjepa.py:542-556) for one mask: x fp32 [B,N,D], idx int32 [B,M] -> [B,M,D] (HIP row gather)."""
    lib = _lib.load()
    B, N, D = x.shape
    out = torch.empty(B, idx.shape[1], D, dtype=torch.float32, device=x.device)
    _lib.check(lib.smbv_gather_rows_idx_f32(x.data_ptr(), out.data_ptr(), idx.data_ptr(), idx.numel(), idx.shape[1], N, D,
                                            torch.cuda.current_stream().cuda_stream), "smbv_gather_rows_idx_f32")
    return out


class _GatherRowsFn(torch.autograd.Function):
    """apply_masks with its autograd: rows of a mask are distinct, so the backward is a plain row scatter into zeros (no atomics)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return gather_rows(x.detach(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, N, D = ctx.shape
        out = torch.zeros(B, N, D, dtype=torch.float32, device=g.device)
        g = g.detach().to(torch.float32).contiguous()
        _lib.check(_lib.load().smbv_scatter_rows_idx_f32(g.data_ptr(), out.data_ptr(), idx.data_ptr(), idx.numel(), idx.shape[1], N, D,
                                                         torch.cuda.current_stream().cuda_stream), "smbv_scatter_rows_idx_f32")
        return out, None


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.load()
        pred, target = pred.contiguous(), target.contiguous()
        part = torch.empty(1024, dtype=torch.float32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(lib.smbv_l1_fwd(pred.data_ptr(), target.data_ptr(), pred.numel(), part.data_ptr(), loss.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "smbv_l1_fwd")
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        d = torch.empty_like(pred)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        _lib.check(lib.smbv_l1_bwd(pred.data_ptr(), target.data_ptr(), pred.numel(), d.data_ptr(), gs.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream), "smbv_l1_bwd")
        return d, None


def l1_loss(pred, target):
    """nn.L1Loss() (run_vjepa.py:108,136) on the HIP path."""
    if not pred.is_cuda:
        raise _lib.SmbvError("smb_vision_amd: l1_loss needs HIP tensors")
    return _L1Fn.apply(pred.float(), target.float())


class MomentumEncoder:
    """run_vjepa.py:87-98: frozen deep copy updated as k <- m*k + (1-m)*q, as ONE multi-tensor HIP launch."""

    def __init__(self, model, momentum=0.99925):
        self.model = model
        self.momentum = momentum
        self._table = None

    def update(self, source_model):
        lib = _lib.load()
        pairs = list(zip(source_model.parameters(), self.model.parameters()))
        sig = tuple((q.data_ptr(), k.data_ptr()) for q, k in pairs)
        if self._table is None or self._table[0] != sig:
            rows = []
            for q, k in pairs:
                for off in range(0, q.numel(), CHUNK):
                    rows.append((k.data_ptr() + 4 * off, q.data_ptr() + 4 * off, 0, 0, min(CHUNK, q.numel() - off), 0))
            arr = np.zeros(len(rows), dtype=np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("group", "<i4")]))
            for i, r in enumerate(rows):
                arr[i] = r
            self._table = (sig, torch.from_numpy(arr.view(np.uint8).copy()).to(pairs[0][0].device), len(rows))
        _, table, n = self._table
        _lib.check(lib.smbv_ema_update(table.data_ptr(), n, float(self.momentum), torch.cuda.current_stream().cuda_stream), "smbv_ema_update")


def make_target_encoder(model: SmbVJEPA2Model) -> SmbVJEPA2Model:
    """VJEPATrainer.__init__ (run_vjepa.py:104-106): frozen deep copy of the online model."""
    pending, model._plumbing_checks = model._plumbing_checks, []   # (events / pinned buffers of queued checks are not part of a copy)
    try:
        t = copy.deepcopy(model)
    finally:
        model._plumbing_checks = pending
    t._gflat = t._structs = t._ws = t._ws_key = None
    for p in t.parameters():
        p.requires_grad = False
    return t


_TARGET_STREAMS = {}


def _target_stream(device):
    key = (device.type, device.index)
    if key not in _TARGET_STREAMS:
        _TARGET_STREAMS[key] = torch.cuda.Stream(device=device)
    return _TARGET_STREAMS[key]


def vjepa_loss(model, target_encoder, momentum_encoder, inputs):
    """VJEPATrainer.compute_loss (run_vjepa.py:110-141): online forward, frozen target forward, L1, EMA update — in that order."""
    # (smbv_needed: compute_loss reads predictor_output from the online model and target_hidden_state from the target encoder, nothing else -
    #  the last encoder block of each then only works on the rows those are made of; SMBV_VJEPA_FULL_ENCODER=1 keeps the unrestricted forward)
    full = os.environ.get("SMBV_VJEPA_FULL_ENCODER", "0") == "1"
    # The frozen target encoder's forward does not depend on the online forward (its weights are last step's EMA): it is enqueued on a second stream
    # so that the two forwards' kernels run side by side and fill each other's partial last rounds (an encoder's attention grid is 2.5 rounds of block
    # slots at S = 20,480 x 12 heads); SMBV_VJEPA_SERIAL_TARGET=1 keeps it behind the online forward on the caller's stream
    pix = inputs["pixel_values_videos"]
    cur = torch.cuda.current_stream(pix.device) if pix.is_cuda else None
    side = None if (cur is None or os.environ.get("SMBV_VJEPA_SERIAL_TARGET", "0") == "1") else _target_stream(pix.device)

    def run_target():
        with torch.no_grad():
            return target_encoder(pixel_values_videos=pix, context_mask=inputs["context_mask"], target_mask=inputs["target_mask"], skip_predictor=True,
                                  smbv_needed=None if full else "target", smbv_plumbing=inputs.get("smbv_plumbing_target")).target_hidden_state

    if side is not None:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            tgt = run_target()
    out = model(pixel_values_videos=inputs["pixel_values_videos"], context_mask=inputs["context_mask"], target_mask=inputs["target_mask"],
                skip_predictor=False, smbv_needed=None if full else "context", smbv_plumbing=inputs.get("smbv_plumbing_context"))
    pred = out.predictor_output.last_hidden_state
    if side is not None:
        cur.wait_stream(side)
        tgt.record_stream(cur)
    else:
        tgt = run_target()
    loss = l1_loss(pred, tgt)
    if momentum_encoder is not None:
        momentum_encoder.update(model)
    return loss, out
