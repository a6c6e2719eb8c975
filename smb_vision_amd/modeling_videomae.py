"""``SmbVideoMAEForPreTraining`` — the MI355X-native drop-in for the model class ``src/run_mim.py`` trains.

Host-side mirror of ``VideoMAEForPreTraining`` (reference spec ``src/models/videomae/modeling_videomae.py:734-908``; the class
``run_mim.py:13-21,345-357`` instantiates): same constructor (``Cls(config: VideoMAEConfig)`` / ``from_pretrained``), same
``forward(pixel_values, bool_masked_pos)`` contract, same output object (``.loss`` first, ``.logits``), same errors, same
state-dict names (pinned-era layout; transformers-5 ``query/key/value.bias`` checkpoints are accepted on
load).  All arithmetic of the step — patch embedding, encoder, decoder, targets, loss, and the whole backward — runs in
libsmbv.so (hand-written HIP for gfx950) through two C calls, ``smbv_mim_forward`` / ``smbv_mim_backward``; PyTorch only
owns device memory, streams and the autograd edge.  There is no CPU or eager fallback: CPU tensors raise.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import logging
import os
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch import nn
from transformers import PreTrainedModel, VideoMAEConfig
from transformers.utils import ModelOutput

from . import _lib
from .masking import split_mask_indices

logger = logging.getLogger(__name__)


def _residual_wanted(config, default: bool = False) -> bool:
    """config.smbv_attention_residual: the attention forward also saves the bf16 residual of its fp32 output and the backward forms delta = rowsum(dO * O)
    from both (smbv_*_cfg.attn_o_resid).  Default: on for V-JEPA, whose encoder attends over all 20,480 near-identical tokens at init scale - the regime in
    which a bf16 O inside delta costs 10-15 % on some query / key gradients; off for MIM (every gradient tensor within 1e-2 of the fp32 oracle at full size
    without it, +0.12 ms per step with it: profiles/r04_resid_ab.txt).  The environment variable SMBV_ATTN_RESID = 0 / 1 overrides both (A/B tools)."""
    env = os.environ.get("SMBV_ATTN_RESID")
    if env in ("0", "1"):
        return env == "1"
    return bool(getattr(config, "smbv_attention_residual", default))


@dataclass
class SmbVideoMAEForPreTrainingOutput(ModelOutput):
    """Same fields as the reference's ``VideoMAEForPreTrainingOutput`` (spec :74-91)."""

    loss: Optional[torch.FloatTensor] = None
    logits: Optional[torch.FloatTensor] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None


def get_sinusoid_encoding_table(n_position: int, d_hid: int) -> torch.Tensor:
    """Fixed sin-cos table, float64 numpy then fp32, ``[1, n_position, d_hid]`` (spec :95-106), vectorised."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.from_numpy(tab.astype(np.float32)).unsqueeze(0)


_LAYER_SUFFIXES = (  # (state-dict suffix, C field, shape code)
    ("layernorm_before.weight", "ln1_w", "w"), ("layernorm_before.bias", "ln1_b", "w"),
    ("attention.attention.query.weight", "q_w", "ww"), ("attention.attention.key.weight", "k_w", "ww"),
    ("attention.attention.value.weight", "v_w", "ww"),
    ("attention.attention.q_bias", "q_b", "w"), ("attention.attention.v_bias", "v_b", "w"),
    ("attention.output.dense.weight", "proj_w", "ww"), ("attention.output.dense.bias", "proj_b", "w"),
    ("layernorm_after.weight", "ln2_w", "w"), ("layernorm_after.bias", "ln2_b", "w"),
    ("intermediate.dense.weight", "fc1_w", "fw"), ("intermediate.dense.bias", "fc1_b", "f"),
    ("output.dense.weight", "fc2_w", "wf"), ("output.dense.bias", "fc2_b", "w"),
)


def param_table(config) -> "list[tuple[str, tuple, tuple]]":
    """[(state-dict name, shape, C-struct path)] in canonical (forward) order."""
    d, ff = config.hidden_size, config.intermediate_size
    dd, dff = config.decoder_hidden_size, config.decoder_intermediate_size
    ts, ps, c = int(config.tubelet_size), _pair(config.patch_size)[0], config.num_channels
    out = [("videomae.embeddings.patch_embeddings.projection.weight", (d, c, ts, ps, ps), ("pe_w",)),
           ("videomae.embeddings.patch_embeddings.projection.bias", (d,), ("pe_b",))]

    def layer(prefix, group, i, w, f):
        shp = {"w": (w,), "ww": (w, w), "fw": (f, w), "f": (f,), "wf": (w, f)}
        for suf, field, code in _LAYER_SUFFIXES:
            out.append((prefix + suf, shp[code], (group, i, field)))

    for i in range(config.num_hidden_layers):
        layer(f"videomae.encoder.layer.{i}.", "enc", i, d, ff)
    out.append(("encoder_to_decoder.weight", (dd, d), ("e2d_w",)))
    out.append(("mask_token", (1, 1, dd), ("mask_token",)))
    for i in range(config.decoder_num_hidden_layers):
        layer(f"decoder.decoder_layers.{i}.", "dec", i, dd, dff)
    K = c * ts * ps * ps
    out += [("decoder.norm.weight", (dd,), ("dnorm_w",)), ("decoder.norm.bias", (dd,), ("dnorm_b",)),
            ("decoder.head.weight", (K, dd), ("head_w",)), ("decoder.head.bias", (K,), ("head_b",))]
    return out


def _pair(x):
    return tuple(x) if isinstance(x, collections.abc.Iterable) else (x, x)


class _Slots(nn.Module):
    """Pure parameter container; the hierarchy only exists to reproduce the reference's state-dict names.  The ``videomae`` node of a
    task model is also callable (``model.videomae(image)``, src/run_inference.py:78-83): its owner installs ``_encode``."""

    def forward(self, pixel_values=None, bool_masked_pos=None, head_mask=None, output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        enc = self.__dict__.get("_encode")
        if enc is None:
            raise TypeError("this container only holds parameters")
        if head_mask is not None or output_attentions:
            raise NotImplementedError("smb_vision_amd: head masks / attention maps are not produced by the fused HIP path")
        out = enc(pixel_values, bool_masked_pos, output_hidden_states=output_hidden_states)
        return out.to_tuple() if return_dict is False else out


def _register(root: nn.Module, dotted: str, p: nn.Parameter):
    mod = root
    parts = dotted.split(".")
    for part in parts[:-1]:
        if part not in mod._modules:
            mod.add_module(part, _Slots())
        mod = mod._modules[part]
    mod.register_parameter(parts[-1], p)


def _set_struct(struct, path, value):
    if len(path) == 1:
        setattr(struct, path[0], value)
    else:
        setattr(getattr(struct, path[0])[path[1]], path[2], value)


class _MIMStepFn(torch.autograd.Function):
    """Autograd edge around the two C calls.  Inputs: every parameter (so autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, model_ref, pixel_values, vis_idx, mask_idx, *params):
        model = model_ref()
        loss, logits, st = model._run_forward(pixel_values, vis_idx, mask_idx)
        ctx.model_ref = model_ref
        ctx.st = st
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # no 109 MB zero tensor for the unused logits gradient
        return loss, logits

    @staticmethod
    def backward(ctx, grad_loss, _grad_logits):
        model = ctx.model_ref()
        if grad_loss is None:  # loss not used by the caller's graph
            return (None,) * (4 + len(model._table))
        grads = model._run_backward(ctx.st, grad_loss)
        ctx.st = None
        return (None, None, None, None, *grads)


class SmbVideoMAEForPreTraining(PreTrainedModel):
    config_class = VideoMAEConfig
    base_model_prefix = "videomae"
    main_input_name = "pixel_values"
    supports_gradient_checkpointing = True
    _supports_sdpa = True
    _supports_flash_attn = True
    _supports_attention_backend = True

    def __init__(self, config: VideoMAEConfig):
        super().__init__(config)
        self.config = config
        if config.num_channels != 1:
            raise ValueError("smb_vision_amd: the MI355X MIM path is built for single-channel volumes (num_channels=1, as src/run_mim.py:318 sets)")
        if getattr(config, "hidden_act", "gelu") != "gelu":
            raise ValueError("smb_vision_amd: only hidden_act='gelu' (exact erf) is implemented")
        if not getattr(config, "qkv_bias", True):
            raise ValueError("smb_vision_amd: qkv_bias=False is not implemented")
        if float(getattr(config, "hidden_dropout_prob", 0.0) or 0.0) != 0.0 or float(getattr(config, "attention_probs_dropout_prob", 0.0) or 0.0) != 0.0:
            # (the reference trains with VideoMAEConfig's defaults, 0.0 / 0.0; the fused kernels have no dropout, and silently ignoring it would change training)
            raise ValueError("smb_vision_amd: hidden_dropout_prob / attention_probs_dropout_prob must be 0 (dropout is not implemented on the HIP path)")
        if not getattr(config, "norm_pix_loss", True):
            # with num_channels != 3 the reference raises here too (spec :869-872)
            raise ValueError("Can't unnormalize non-RGB images. Consider setting config.norm_pix_loss to False.")
        self._table = param_table(config)
        for name, shape, _ in self._table:
            _register(self, name, nn.Parameter(torch.empty(shape)))
        self.image_size = _pair(config.image_size)
        self.patch_size = _pair(config.patch_size)
        if self.patch_size[0] != self.patch_size[1]:
            raise ValueError("smb_vision_amd: square patches only")
        self.tubelet_size = int(config.tubelet_size)
        self.num_patches = (self.image_size[1] // self.patch_size[1]) * (self.image_size[0] // self.patch_size[0]) * (
            config.num_frames // self.tubelet_size)
        self.gradient_checkpointing = False  # see _use_recompute
        # non-persistent, device-resident state
        self._pos = None          # (pos_enc, pos_dec) on the parameter device
        self._gflat = None        # flat fp32 gradient arena; param.grad are views into it
        self._structs = None      # (ptr signature, MimTensors params, MimTensors grads)
        self._ws = None           # workspace tensor
        self._ws_key = None
        self._n_masked = None     # cached token split (validated lazily, no per-step host sync)
        self._pending_check = None
        self._fwd_count = 0
        self.dp = None            # optional DataParallelEngine (smb_vision_amd.dp)
        self.reset_parameters()
        self.post_init()

        def _encode(pixel_values, bool_masked_pos=None, output_hidden_states=None, _ref=weakref.ref(self)):
            from .modeling_videomae_encoder import encode_with
            return encode_with(_ref(), pixel_values, bool_masked_pos, output_hidden_states)
        object.__setattr__(self._modules["videomae"], "_encode", _encode)

    # ------------------------------------------------------------------------------------------------------------
    # parameters
    # ------------------------------------------------------------------------------------------------------------
    def reset_parameters(self):
        """_init_weights semantics of the reference (spec :495-505, :742, :247-248)."""
        std = self.config.initializer_range
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("weight") and p.dim() >= 2:
                    p.normal_(mean=0.0, std=std)
                elif "layernorm" in name or name.startswith("decoder.norm"):
                    p.fill_(1.0) if name.endswith("weight") else p.zero_()
                else:
                    p.zero_()

    def _init_weights(self, module):
        # parameters live in _Slots containers and are initialised by reset_parameters(); from_pretrained calls this only for
        # containers whose parameters were missing from the checkpoint
        if not isinstance(module, _Slots) and module is not self:
            return
        names = {id(p): n for n, p in self.named_parameters()}
        std = self.config.initializer_range
        with torch.no_grad():
            for _, p in module.named_parameters(recurse=False):
                if getattr(p, "_is_hf_initialized", False):
                    continue
                n = names.get(id(p), "")
                if n.endswith("weight") and p.dim() >= 2:
                    p.normal_(mean=0.0, std=std)
                elif ("layernorm" in n or n.startswith("decoder.norm")) and n.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()

    def _set_gradient_checkpointing(self, enable: bool = True, gradient_checkpointing_func=None):
        self.gradient_checkpointing = enable

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        # HF Trainer calls this with --gradient_checkpointing (trainer.py:1401); see _use_recompute for what it switches on.
        self.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        """As ``PreTrainedModel.from_pretrained`` (``src/run_mim.py:345-355``), additionally accepting checkpoints written by
        transformers >= 5 whose attention has ``query/key/value.bias`` instead of ``q_bias/v_bias``."""
        if "key_mapping" not in kwargs:
            kwargs["key_mapping"] = dict(_HF5_KEY_MAPPING)
        return super().from_pretrained(pretrained_model_name_or_path, *model_args, **kwargs)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        convert_state_dict_keys(state_dict, prefix)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = dict(state_dict)
        convert_state_dict_keys(sd, "")
        return super().load_state_dict(sd, strict=strict, assign=assign)

    # ------------------------------------------------------------------------------------------------------------
    # device state
    # ------------------------------------------------------------------------------------------------------------
    def _cfg_struct(self, B, n_masked):
        c = self.config
        s = _lib.MimCfg()
        s.B, s.T, s.H, s.W = B, c.num_frames, self.image_size[0], self.image_size[1]
        s.ps, s.ts = self.patch_size[0], self.tubelet_size
        s.N, s.Nvis, s.Nmask = self.num_patches, self.num_patches - n_masked, n_masked
        s.d, s.heads, s.ff, s.layers = c.hidden_size, c.num_attention_heads, c.intermediate_size, c.num_hidden_layers
        s.dd, s.dheads, s.dff, s.dlayers = (c.decoder_hidden_size, c.decoder_num_attention_heads, c.decoder_intermediate_size,
                                            c.decoder_num_hidden_layers)
        s.ln_eps = c.layer_norm_eps
        s.dec_norm_eps = 1e-5
        s.recompute = int(self._use_recompute(s))
        # True / "forward": the attention FORWARD's QK^T / PV on the fp8 MFMA (what the flag has meant since round 2); "all": also the backward's recomputed
        # S / dP products (round 4) - explicit, because it changes the backward's numerics (dQ 5.8e-2 rms from exact instead of 3e-3, tests/test_fp8_gpu.py) and
        # switches the attention residual off; training CONVERGENCE with it is unpinned (per-step gradient bounds against the oracle are what is tested)
        f8 = getattr(c, "smbv_attention_fp8", False)
        if f8 not in (False, None, 0, True, "forward", "all"):
            raise ValueError(f"config.smbv_attention_fp8 must be False, True / 'forward' or 'all', got {f8!r}")
        s.attn_fp8 = 0 if not f8 else (3 if f8 == "all" else 1)
        if s.attn_fp8 and getattr(self, "_fp8_mode_logged", None) != s.attn_fp8:
            self._fp8_mode_logged = s.attn_fp8
            logger.info("smb_vision_amd: fp8 attention %s", "forward + backward recomputation of S / dP (smbv_attention_fp8='all')" if s.attn_fp8 == 3
                        else "forward only (smbv_attention_fp8=True / 'forward')")
        # delta = rowsum(dO * O) from the attention output to 16 significant bits (smbv_mim_cfg.attn_o_resid; one more bf16 activation per block)
        s.attn_o_resid = int(_residual_wanted(c) and not s.attn_fp8)
        return s

    def _use_recompute(self, cfg) -> bool:
        """``gradient_checkpointing_enable()`` (HF Trainer ``--gradient_checkpointing``, which the reference's scripts always pass) asks for the
        per-layer recompute of spec :459-465.  Results are bit-identical either way; recomputing costs one more forward.  Policy
        ``config.smbv_recompute``: "always" / "never" / "auto" (default): recompute only when keeping every block's activations would take more
        than 60 % of the device's memory - on a 288 GB MI355X ViT-base / ViT-large at batch 1-4 stay resident."""
        if not self.gradient_checkpointing:
            return False
        policy = getattr(self.config, "smbv_recompute", "auto")
        if policy in ("always", True):
            return True
        if policy in ("never", False):
            return False
        key = (cfg.B, cfg.Nmask)
        cache = self.__dict__.setdefault("_recompute_auto", {})
        if key not in cache:
            cfg.recompute = 0
            need = _lib.load().smbv_mim_workspace_bytes(C.byref(cfg))
            total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory if torch.cuda.is_available() else 0
            cache[key] = bool(total and need > 0.6 * total)
        return cache[key]

    def _ordered_params(self):
        sd = dict(self.named_parameters())
        return [sd[name] for name, _, _ in self._table]

    def _ensure_device_state(self, device):
        params = self._ordered_params()
        for p in params:
            if p.device != device or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SmbvError("smb_vision_amd: parameters must be contiguous fp32 tensors on the input's HIP device "
                                     f"(found {p.dtype} on {p.device}); mixed precision is internal to the kernels")
        if self._pos is None or self._pos[0].device != device:
            self._pos = (get_sinusoid_encoding_table(self.num_patches, self.config.hidden_size)[0].to(device).contiguous(),
                         get_sinusoid_encoding_table(self.num_patches, self.config.decoder_hidden_size)[0].to(device).contiguous())
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
            ps, gs = _lib.MimTensors(), _lib.MimTensors()
            gbase = self._gflat.data_ptr()
            for p, off, (_, _, path) in zip(params, self._goffs, self._table):
                _set_struct(ps, path, p.data_ptr())
                _set_struct(gs, path, gbase + 4 * off)
            self._structs = (sig, ps, gs)
        return params

    def grad_views(self):
        """Fresh per-parameter views into the flat gradient arena (canonical order)."""
        return [self._gflat[off:off + p.numel()].view(p.shape) for p, off in zip(self._ordered_params(), self._goffs)]

    def flat_grads(self) -> torch.Tensor:
        return self._gflat

    def grad_group_ranges(self):
        """Element ranges of the flat gradient arena in the order smbv_mim_backward completes them (its event order):
        head+decoder.norm, decoder layers (last first), mask_token+encoder_to_decoder, encoder layers (last first), patch embed."""
        names = [n for n, _, _ in self._table]
        sizes = [(p.numel() + 63) // 64 * 64 for p in self._ordered_params()]
        offs = self._goffs
        L, DL = self.config.num_hidden_layers, self.config.decoder_num_hidden_layers

        def rng(pred):
            idx = [i for i, n in enumerate(names) if pred(n)]
            return (offs[idx[0]], offs[idx[-1]] + sizes[idx[-1]])

        out = [rng(lambda n: n.startswith("decoder.norm") or n.startswith("decoder.head"))]
        for i in reversed(range(DL)):
            out.append(rng(lambda n, i=i: n.startswith(f"decoder.decoder_layers.{i}.")))
        out.append(rng(lambda n: n in ("encoder_to_decoder.weight", "mask_token")))
        for i in reversed(range(L)):
            out.append(rng(lambda n, i=i: n.startswith(f"videomae.encoder.layer.{i}.")))
        out.append(rng(lambda n: n.startswith("videomae.embeddings.")))
        return out

    # ------------------------------------------------------------------------------------------------------------
    # the step
    # ------------------------------------------------------------------------------------------------------------
    def _token_split(self, bool_masked_pos, num_masked):
        N = bool_masked_pos.shape[1]
        if num_masked is None:
            if self._pending_check is not None:
                ev, host = self._pending_check
                if ev.query():
                    self._pending_check = None
                    if int(host[0]) != 0:
                        raise ValueError("bool_masked_pos: the number of masked tokens changed between steps or differs across the batch")
            if self._n_masked is None:
                counts = bool_masked_pos.sum(dim=1)
                self._n_masked = int(counts[0])  # one host sync, first step only
                if bool((counts != self._n_masked).any()):
                    raise ValueError("bool_masked_pos: all samples in a batch must mask the same number of tokens")
            elif self._pending_check is None:
                host = torch.empty(1, dtype=torch.int32).pin_memory()
                bad = (bool_masked_pos.sum(dim=1) != self._n_masked).any().to(torch.int32).reshape(1)
                host.copy_(bad, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending_check = (ev, host)
            num_masked = self._n_masked
        if not 0 < num_masked < N:
            raise ValueError(f"bool_masked_pos must mask between 1 and {N - 1} tokens, got {num_masked}")
        return num_masked

    def _run_forward(self, pixel_values, vis_idx, mask_idx):
        lib = _lib.load()
        dev = pixel_values.device
        B = pixel_values.shape[0]
        n_masked = mask_idx.shape[1]
        cfg = self._cfg_struct(B, n_masked)
        need = lib.smbv_mim_workspace_bytes(C.byref(cfg))
        if need == 0:
            raise ValueError("smbv_mim_workspace_bytes: " + _lib.last_error())
        key = (B, n_masked, dev, cfg.recompute, cfg.attn_fp8, cfg.attn_o_resid)
        if self._ws is None or self._ws_key != key or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws_key = key
        K = self.tubelet_size * self.patch_size[0] ** 2
        loss = torch.empty((), dtype=torch.float32, device=dev)
        logits = torch.empty(B, n_masked, K, dtype=torch.bfloat16, device=dev)
        io = _lib.MimIO()
        io.pixel_values, io.vis_idx, io.mask_idx = pixel_values.data_ptr(), vis_idx.data_ptr(), mask_idx.data_ptr()
        io.pos_enc, io.pos_dec = self._pos[0].data_ptr(), self._pos[1].data_ptr()
        io.loss, io.logits, io.grad_scale = loss.data_ptr(), logits.data_ptr(), None
        _, ps, gs = self._structs
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.smbv_mim_forward(C.byref(cfg), C.byref(ps), C.byref(io), self._ws.data_ptr(), self._ws.numel(), stream),
                   "smbv_mim_forward")
        self._fwd_count += 1
        st = dict(cfg=cfg, io=io, step=self._fwd_count, keep=(pixel_values, vis_idx, mask_idx, logits, loss, self._ws))
        return loss, logits, st

    def _run_backward(self, st, grad_loss):
        lib = _lib.load()
        if st["step"] != self._fwd_count:
            raise _lib.SmbvError("smb_vision_amd: backward() after a later forward() reused the step workspace "
                                 "(one forward/backward pair at a time per model)")
        _lib.detach_aliased_grads(self._ordered_params(), self._gflat, self._goffs, self)  # accumulation / zero_grad(set_to_none=False)
        gscale = grad_loss.detach().to(torch.float32).reshape(1).contiguous()
        io = st["io"]
        io.grad_scale = gscale.data_ptr()
        _, ps, gs = self._structs
        stream = torch.cuda.current_stream().cuda_stream
        events, n_ev = None, 0
        if self.dp is not None:
            self.dp.prepare(self)  # per-group events exist before the first backward records them
            events, n_ev = self.dp.event_array()
        _lib.check(lib.smbv_mim_backward(C.byref(st["cfg"]), C.byref(ps), C.byref(gs), C.byref(io), self._ws.data_ptr(), self._ws.numel(),
                                         events, n_ev, stream), "smbv_mim_backward")
        if self.dp is not None:
            self.dp.allreduce_after_backward(self)
        return self.grad_views()

    def forward(self, pixel_values: torch.Tensor, bool_masked_pos: torch.Tensor, head_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, num_masked: Optional[int] = None, visible_index: Optional[torch.Tensor] = None,
                masked_index: Optional[torch.Tensor] = None, **kwargs):
        """``VideoMAEForPreTraining.forward`` (spec :753-908).  ``visible_index`` / ``masked_index`` (optional, int32 [B, N-M] / [B, M], ascending - what
        ``masking.split_mask_indices_host`` builds in the collate function from the mask it just generated) spare the per-step device sort that
        otherwise derives them from ``bool_masked_pos``; they MUST describe the same partition."""
        if head_mask is not None or output_attentions or (output_attentions is None and getattr(self.config, "output_attentions", False)):
            raise NotImplementedError("smb_vision_amd: head masks / attention maps ([B, H, N, N]) are never materialised by the fused HIP path")
        want_hidden = bool(output_hidden_states if output_hidden_states is not None else getattr(self.config, "output_hidden_states", False))  # spec :790-793
        if pixel_values.dim() != 5:
            raise ValueError("pixel_values must be [batch, frames, channels, height, width]")
        B, T, Cc, H, W = pixel_values.shape
        if Cc != self.config.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
        if H != self.image_size[0] or W != self.image_size[1]:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({self.image_size[0]}*{self.image_size[1]}).")
        if T != self.config.num_frames:
            raise ValueError(f"Input depth ({T}) doesn't match model ({self.config.num_frames}).")
        if bool_masked_pos is None:
            raise ValueError("One must provided a boolean mask ")
        if not pixel_values.is_cuda:
            raise _lib.SmbvError("smb_vision_amd: pixel_values must live on a HIP device; there is no CPU fallback for the MIM hot path")
        _lib.load()
        dev = pixel_values.device
        with torch.cuda.device(dev):
            pixel_values = pixel_values.to(torch.float32).contiguous()
            bool_masked_pos = bool_masked_pos.to(device=dev, dtype=torch.bool)
            if bool_masked_pos.shape != (B, self.num_patches):
                raise ValueError(f"bool_masked_pos must be [{B}, {self.num_patches}], got {tuple(bool_masked_pos.shape)}")
            params = self._ensure_device_state(dev)
            if visible_index is not None and masked_index is not None:
                vis_idx = visible_index.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous()
                mask_idx = masked_index.to(device=dev, dtype=torch.int32, non_blocking=True).contiguous()
                n_masked = mask_idx.shape[1]
                if vis_idx.shape != (B, self.num_patches - n_masked) or mask_idx.shape[0] != B or not 0 < n_masked < self.num_patches:
                    raise ValueError(f"visible_index / masked_index must be [{B}, N-M] / [{B}, M] with N = {self.num_patches}")
            else:
                n_masked = self._token_split(bool_masked_pos, num_masked)
                vis_idx, mask_idx = split_mask_indices(bool_masked_pos, n_masked)
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                loss, logits = _MIMStepFn.apply(weakref.ref(self), pixel_values, vis_idx, mask_idx, *params)
            else:
                loss, logits, _ = self._run_forward(pixel_values, vis_idx, mask_idx)
            hidden = self._encoder_hidden_states(B, n_masked) if want_hidden else None
        if return_dict is False:
            return (loss, logits) + ((hidden,) if want_hidden else ())
        return SmbVideoMAEForPreTrainingOutput(loss=loss, logits=logits, hidden_states=hidden)

    def _encoder_hidden_states(self, B, n_masked):
        """``hidden_states`` of the reference's output (spec :900-907: the ENCODER's, over the visible tokens - the embedding output and every layer's output),
        copied out of the step workspace the forward just filled (``smbv_mim_hidden_states``; every mode keeps them).  Detached, fp32."""
        lib = _lib.load()
        cfg = self._cfg_struct(B, n_masked)
        n = cfg.layers + 1
        ptrs = (C.c_void_p * n)()
        _lib.check(lib.smbv_mim_hidden_states(C.byref(cfg), self._ws.data_ptr(), self._ws.numel(), ptrs, n), "smbv_mim_hidden_states")
        base, rows = self._ws.data_ptr(), self.num_patches - n_masked
        nb = B * rows * cfg.d * 4
        return tuple(self._ws[int(ptrs[l]) - base:int(ptrs[l]) - base + nb].view(torch.float32).view(B, rows, cfg.d).clone() for l in range(n))


_HF5_KEY_MAPPING = {
    r"attention\.attention\.query\.bias$": "attention.attention.q_bias",
    r"attention\.attention\.value\.bias$": "attention.attention.v_bias",
}


def convert_state_dict_keys(sd: dict, prefix: str = ""):
    """Accept transformers-5 checkpoints: ``query.bias -> q_bias``, ``value.bias -> v_bias``; a ``key.bias`` adds the same
    constant to every score of a query row, which softmax ignores, so it is dropped (exact)."""
    for k in list(sd.keys()):
        if not k.startswith(prefix):
            continue
        if k.endswith("attention.attention.query.bias"):
            sd[k[: -len("query.bias")] + "q_bias"] = sd.pop(k)
        elif k.endswith("attention.attention.value.bias"):
            sd[k[: -len("value.bias")] + "v_bias"] = sd.pop(k)
        elif k.endswith("attention.attention.key.bias"):
            sd.pop(k)
    return sd
