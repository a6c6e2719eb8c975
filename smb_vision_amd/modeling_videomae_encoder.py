"""Encoder-only forwards on the MI355X path: the embedding export and the classification forward of the
reference, same constructor / ``from_pretrained`` / ``forward`` signature / output objects / state-dict names.

* ``SmbVideoMAEModel``  - ``VideoMAEModel`` (``src/models/videomae/modeling_videomae.py:509-655``): ``forward(pixel_values,
  bool_masked_pos=None)`` -> ``BaseModelOutput(last_hidden_state [B, N or N_visible, d])``; final ``layernorm`` only when
  ``config.use_mean_pooling`` is False (spec :517-520).
* ``SmbVideoMAEForVideoClassification`` - spec :917-1010 with ``use_mean_pooling=True``: mean over tokens -> ``fc_norm`` ->
  (concat ``additional_features``) -> ``classifier``; losses as the reference computes them.
* ``encode_with(model, pixel_values, bool_masked_pos=None)`` - ``model.videomae(image)`` of ``src/run_inference.py:78-83`` for a
  ``SmbVideoMAEForPreTraining`` (it is also installed as ``model.videomae.forward``).

Under ``torch.no_grad()`` one C call (``smbv_encoder_forward``, one shared activation set); with autograd enabled the layers keep their
activations and ``smbv_encoder_backward`` returns the encoder's parameter gradients (fine-tuning, ``src/run_classification.py``); the
``[B, d]`` head (mean pool, ``fc_norm``, classifier, loss) is ordinary tensor code on the device.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch
from torch import nn
from transformers import PreTrainedModel, VideoMAEConfig
from transformers.modeling_outputs import BaseModelOutput, ImageClassifierOutput

from . import _lib
from .modeling_videomae import _residual_wanted
from .masking import split_mask_indices
from .modeling_videomae import (_HF5_KEY_MAPPING, _LAYER_SUFFIXES, _Slots, _pair, _register, convert_state_dict_keys,
                                get_sinusoid_encoding_table)


def encoder_table(config, prefix: str, final_ln: bool):
    """[(state-dict name, shape, C path)] of the encoder under ``prefix`` ('' for VideoMAEModel, 'videomae.' inside the task models)."""
    d, ff = config.hidden_size, config.intermediate_size
    ts, ps, c = int(config.tubelet_size), _pair(config.patch_size)[0], config.num_channels
    out = [(prefix + "embeddings.patch_embeddings.projection.weight", (d, c, ts, ps, ps), ("pe_w",)),
           (prefix + "embeddings.patch_embeddings.projection.bias", (d,), ("pe_b",))]
    shp = {"w": (d,), "ww": (d, d), "fw": (ff, d), "f": (ff,), "wf": (d, ff)}
    for i in range(config.num_hidden_layers):
        for suf, field, code in _LAYER_SUFFIXES:
            out.append((f"{prefix}encoder.layer.{i}.{suf}", shp[code], ("enc", i, field)))
    if final_ln:
        out += [(prefix + "layernorm.weight", (d,), ("lnf_w",)), (prefix + "layernorm.bias", (d,), ("lnf_b",))]
    return out


class _EncoderFn(torch.autograd.Function):
    """Autograd edge around smbv_encoder_forward (train mode) / smbv_encoder_backward.  Inputs: every encoder parameter."""

    @staticmethod
    def forward(ctx, runner, pix, sel, nsel, *params):
        out, st = runner._run(pix, sel, nsel, params, train=True)
        ctx.runner, ctx.st = runner, st
        runner._last_st = dict(cfg=st["cfg"])   # (what hidden_states() needs; not the tensors)
        return out

    @staticmethod
    def backward(ctx, d_out):
        grads = ctx.runner._run_backward(ctx.st, d_out)
        ctx.st = None
        return (None, None, None, None, *grads)


class _EncoderRunner:
    """Device state + the C call, shared by every class that owns encoder parameters under some prefix."""

    def __init__(self, owner: nn.Module, config, prefix: str, final_ln: bool):
        self.owner = weakref.ref(owner)
        self.config, self.prefix, self.final_ln = config, prefix, final_ln
        self.table = encoder_table(config, prefix, final_ln)
        self.image_size = _pair(config.image_size)
        self.ps = _pair(config.patch_size)[0]
        self.ts = int(config.tubelet_size)
        self.N = (self.image_size[0] // self.ps) * (self.image_size[1] // self.ps) * (config.num_frames // self.ts)
        self._pos = None
        self._ws = None
        self._structs = None

    def _params(self):
        sd = dict(self.owner().named_parameters())
        return [sd[n] for n, _, _ in self.table]

    def _tensors(self, params):
        sig = tuple(p.data_ptr() for p in params)
        if self._structs is None or self._structs[0] != sig:
            t = _lib.EncoderTensors()
            for (name, _, path), p in zip(self.table, params):
                if not p.is_contiguous() or p.dtype != torch.float32:
                    raise _lib.SmbvError(f"smb_vision_amd: parameter {name} must be contiguous fp32")
                if len(path) == 1:
                    setattr(t, path[0], p.data_ptr())
                else:
                    setattr(getattr(t, path[0])[path[1]], path[2], p.data_ptr())
            self._structs = (sig, t)
        return self._structs[1]

    def _grad_arena(self, params, dev):
        """One flat fp32 arena for the encoder's gradients (256-byte aligned slots), and the C struct that points into it."""
        sig = tuple(p.data_ptr() for p in params)
        if getattr(self, "_garena", None) is None or self._garena[0] != sig:
            offs, off = [], 0
            for p in params:
                offs.append(off)
                off += (p.numel() + 63) & ~63
            flat = torch.zeros(off, dtype=torch.float32, device=dev)
            t = _lib.EncoderTensors()
            views = []
            for (name, _, path), p, o in zip(self.table, params, offs):
                v = flat[o:o + p.numel()].view(p.shape)
                views.append(v)
                if len(path) == 1:
                    setattr(t, path[0], v.data_ptr())
                else:
                    setattr(getattr(t, path[0])[path[1]], path[2], v.data_ptr())
            self._garena = (sig, flat, t, views)
        return self._garena

    def _run(self, pix, sel, nsel, params, train):
        c = self.config
        dev = pix.device
        lib = _lib.load()
        B, T, _, H, W = pix.shape
        cfg = _lib.EncoderCfg()
        cfg.B, cfg.T, cfg.H, cfg.W, cfg.ps, cfg.ts = B, T, H, W, self.ps, self.ts
        cfg.N, cfg.Nsel = self.N, nsel
        cfg.d, cfg.heads, cfg.ff, cfg.layers = c.hidden_size, c.num_attention_heads, c.intermediate_size, c.num_hidden_layers
        cfg.ln_eps, cfg.final_ln, cfg.final_ln_eps, cfg.train = c.layer_norm_eps, int(self.final_ln), c.layer_norm_eps, int(train)
        cfg.attn_o_resid = int(bool(train and _residual_wanted(c)))
        need = lib.smbv_encoder_workspace_bytes(C.byref(cfg))
        if need == 0:
            raise _lib.SmbvError("smbv_encoder_workspace_bytes: " + _lib.last_error())
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(B, nsel, c.hidden_size, dtype=torch.float32, device=dev)
        io = _lib.EncoderIO()
        io.pixel_values, io.sel_idx, io.pos_enc, io.last_hidden_state = pix.data_ptr(), sel.data_ptr(), self._pos.data_ptr(), out.data_ptr()
        tens = self._tensors(list(params))
        _lib.check(lib.smbv_encoder_forward(C.byref(cfg), C.byref(tens), C.byref(io), self._ws.data_ptr(), self._ws.numel(),
                                            torch.cuda.current_stream().cuda_stream), "smbv_encoder_forward")
        return out, dict(cfg=cfg, io=io, tens=tens, keep=(pix, sel, out), params=list(params))

    def hidden_states(self, st):
        """``hidden_states`` of the reference's encoder (spec :452-457, 478-483: the embedding output and every layer's output, not the final LayerNorm's)
        as copies out of the workspace of the train-mode forward ``st`` came from.  Detached: the fused backward differentiates ``last_hidden_state`` only."""
        lib = _lib.load()
        cfg = st["cfg"]
        n = cfg.layers + 1
        ptrs = (C.c_void_p * n)()
        _lib.check(lib.smbv_encoder_hidden_states(C.byref(cfg), self._ws.data_ptr(), self._ws.numel(), ptrs, n), "smbv_encoder_hidden_states")
        base, nb = self._ws.data_ptr(), cfg.B * cfg.Nsel * cfg.d * 4
        out = []
        for l in range(n):
            off = int(ptrs[l]) - base
            out.append(self._ws[off:off + nb].view(torch.float32).view(cfg.B, cfg.Nsel, cfg.d).clone())
        return tuple(out)

    def _run_backward(self, st, d_out):
        lib = _lib.load()
        dev = d_out.device
        with torch.cuda.device(dev):
            _, flat, gt, views = self._grad_arena(st["params"], dev)
            d = d_out.to(torch.float32).contiguous()
            _lib.check(lib.smbv_encoder_backward(C.byref(st["cfg"]), C.byref(st["tens"]), C.byref(gt), C.byref(st["io"]), d.data_ptr(), self._ws.data_ptr(),
                                                 self._ws.numel(), torch.cuda.current_stream().cuda_stream), "smbv_encoder_backward")
        return [v.clone() for v in views]  # autograd accumulates into .grad; the arena is rewritten by the next backward

    def encode(self, pixel_values: torch.Tensor, bool_masked_pos: Optional[torch.Tensor] = None, output_hidden_states: bool = False):
        """``last_hidden_state``; with ``output_hidden_states`` the pair (last_hidden_state, hidden_states tuple)."""
        c = self.config
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        if not pixel_values.is_cuda:
            raise _lib.SmbvError("smb_vision_amd: pixel_values must live on a HIP device; there is no CPU fallback")
        B, T, Cc, H, W = pixel_values.shape
        if Cc != c.num_channels:
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
        if H != self.image_size[0] or W != self.image_size[1]:
            raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({self.image_size[0]}*{self.image_size[1]}).")
        if T != c.num_frames:
            raise ValueError(f"Input has {T} frames, the model is configured for {c.num_frames}.")
        dev = pixel_values.device
        _lib.load()
        with torch.cuda.device(dev):
            params = self._params()
            if any(p.device != dev for p in params):
                raise _lib.SmbvError("smb_vision_amd: parameters and pixel_values must be on the same HIP device")
            if self._pos is None or self._pos.device != dev:
                self._pos = get_sinusoid_encoding_table(self.N, c.hidden_size)[0].to(dev).contiguous()
            pix = pixel_values.to(torch.float32).contiguous()
            if bool_masked_pos is None:
                nsel = self.N
                sel = torch.arange(self.N, device=dev, dtype=torch.int32).unsqueeze(0).repeat(B, 1).contiguous()
            else:
                m = bool_masked_pos.to(device=dev, dtype=torch.bool)
                counts = m.sum(1)
                n_masked = int(counts[0])  # (host sync: not the pre-training hot path)
                if bool((counts != n_masked).any()):
                    raise ValueError("every volume of the batch must mask the same number of tokens")
                nsel = self.N - n_masked
                sel, _ = split_mask_indices(m, n_masked)
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                out = _EncoderFn.apply(self, pix, sel, nsel, *params)  # fine-tuning: activations kept, backward available
                if output_hidden_states:
                    return out, self.hidden_states(self._last_st)
                return out
            # (the inference forward shares one activation set between its layers: hidden states need the per-layer form)
            out, st = self._run(pix, sel, nsel, params, train=bool(output_hidden_states))
            if output_hidden_states:
                return out, self.hidden_states(st)
        return out


def _check_config(config):
    if config.num_channels != 1:
        raise ValueError("smb_vision_amd: the MI355X path is built for single-channel volumes (num_channels=1)")
    if getattr(config, "hidden_act", "gelu") != "gelu":
        raise ValueError("smb_vision_amd: only hidden_act='gelu' (exact erf) is implemented")
    if not getattr(config, "qkv_bias", True):
        raise ValueError("smb_vision_amd: qkv_bias=False is not implemented")
    if float(getattr(config, "hidden_dropout_prob", 0.0) or 0.0) != 0.0 or float(getattr(config, "attention_probs_dropout_prob", 0.0) or 0.0) != 0.0:
        raise ValueError("smb_vision_amd: hidden_dropout_prob / attention_probs_dropout_prob must be 0 (dropout is not implemented on the HIP path)")


class _EncoderModelBase(PreTrainedModel):
    config_class = VideoMAEConfig
    base_model_prefix = "videomae"
    main_input_name = "pixel_values"
    _enc_prefix = ""

    def _make(self, config, final_ln, extra=()):
        _check_config(config)
        self._runner = _EncoderRunner(self, config, self._enc_prefix, final_ln)
        for name, shape, _ in self._runner.table:
            _register(self, name, nn.Parameter(torch.empty(shape)))
        for name, shape in extra:
            _register(self, name, nn.Parameter(torch.empty(shape)))
        self.reset_parameters()
        self.post_init()

    def reset_parameters(self):
        """_init_weights semantics of the reference (spec :495-505)."""
        std = self.config.initializer_range
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("weight") and p.dim() >= 2:
                    p.normal_(mean=0.0, std=std)
                elif ("layernorm" in name or "fc_norm" in name) and name.endswith("weight"):
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
                if n.endswith("weight") and p.dim() >= 2:
                    p.normal_(mean=0.0, std=std)
                elif ("layernorm" in n or "fc_norm" in n) and n.endswith("weight"):
                    p.fill_(1.0)
                else:
                    p.zero_()

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        if "key_mapping" not in kwargs:
            kwargs["key_mapping"] = dict(_HF5_KEY_MAPPING)
        return super().from_pretrained(pretrained_model_name_or_path, *model_args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = dict(state_dict)
        convert_state_dict_keys(sd, "")
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def _reject(self, head_mask, output_attentions, output_hidden_states=None):
        """Resolves ``output_hidden_states`` against the config as the reference does (spec :596-599); head masks and attention maps ([B, H, N, N]: 10 GB per
        layer at 20,480 tokens) are never materialised by the fused path."""
        if head_mask is not None or output_attentions or (output_attentions is None and getattr(self.config, "output_attentions", False)):
            raise NotImplementedError("smb_vision_amd: head masks / attention maps are not produced by the fused HIP path")
        return bool(output_hidden_states if output_hidden_states is not None else getattr(self.config, "output_hidden_states", False))


class SmbVideoMAEModel(_EncoderModelBase):
    def __init__(self, config: VideoMAEConfig):
        super().__init__(config)
        self._make(config, final_ln=not getattr(config, "use_mean_pooling", True))

    def get_input_embeddings(self):
        """``VideoMAEModel.get_input_embeddings`` (spec :525-526): the patch-embedding node (here the container of ``projection.{weight,bias}``)."""
        return self._modules["embeddings"]._modules["patch_embeddings"]

    def forward(self, pixel_values: torch.Tensor = None, bool_masked_pos: Optional[torch.Tensor] = None, head_mask=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, **kwargs):
        want = self._reject(head_mask, output_attentions, output_hidden_states)
        seq, hidden = self._runner.encode(pixel_values, bool_masked_pos, True) if want else (self._runner.encode(pixel_values, bool_masked_pos), None)
        if return_dict is False:
            return (seq,) + ((hidden,) if want else ())
        return BaseModelOutput(last_hidden_state=seq, hidden_states=hidden, attentions=None)


class SmbVideoMAEForVideoClassification(_EncoderModelBase):
    _enc_prefix = "videomae."

    def __init__(self, config: VideoMAEConfig):
        super().__init__(config)
        if not getattr(config, "use_mean_pooling", True):
            raise NotImplementedError("smb_vision_amd: the classification forward is built for use_mean_pooling=True (the reference default)")
        self.num_labels = config.num_labels
        extra_in = int(getattr(config, "additional_features_size", 0) or 0)
        extra = [("fc_norm.weight", (config.hidden_size,)), ("fc_norm.bias", (config.hidden_size,))]
        if config.num_labels > 0:
            extra += [("classifier.weight", (config.num_labels, config.hidden_size + extra_in)), ("classifier.bias", (config.num_labels,))]
        self._make(config, final_ln=False, extra=extra)

        def _encode(pixel_values, bool_masked_pos=None, output_hidden_states=None, _ref=weakref.ref(self)):
            if output_hidden_states:
                seq, hidden = _ref()._runner.encode(pixel_values, bool_masked_pos, True)
                return BaseModelOutput(last_hidden_state=seq, hidden_states=hidden)
            return BaseModelOutput(last_hidden_state=_ref()._runner.encode(pixel_values, bool_masked_pos))
        object.__setattr__(self._modules["videomae"], "_encode", _encode)

    def forward(self, pixel_values: Optional[torch.Tensor] = None, additional_features: Optional[torch.Tensor] = None, head_mask=None,
                labels: Optional[torch.Tensor] = None, output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        want = self._reject(head_mask, output_attentions, output_hidden_states)
        seq, hidden = self._runner.encode(pixel_values, None, True) if want else (self._runner.encode(pixel_values), None)
        sd = dict(self.named_parameters())
        # the head works on [B, d]: plain tensor ops (fc_norm uses torch's default eps, spec :925)
        pooled = torch.nn.functional.layer_norm(seq.mean(1), (self.config.hidden_size,), sd["fc_norm.weight"], sd["fc_norm.bias"], 1e-5)
        if additional_features is not None:
            if not hasattr(self.config, "additional_features_size"):
                raise ValueError("Model config must have additional_features_size set when using additional_features")
            if additional_features.shape[-1] != self.config.additional_features_size:
                raise ValueError(f"Expected additional_features of size {self.config.additional_features_size}, got {additional_features.shape[-1]}")
            pooled = torch.cat([pooled, additional_features.to(pooled)], dim=-1)
        logits = torch.nn.functional.linear(pooled, sd["classifier.weight"], sd["classifier.bias"]) if self.num_labels > 0 else pooled
        loss = None
        if labels is not None:
            if self.config.problem_type is None:
                if self.num_labels == 1:
                    self.config.problem_type = "regression"
                elif self.num_labels > 1 and labels.dtype in (torch.long, torch.int):
                    self.config.problem_type = "single_label_classification"
                else:
                    self.config.problem_type = "multi_label_classification"
            if self.config.problem_type == "regression":
                loss = nn.functional.mse_loss(logits.squeeze(), labels.squeeze()) if self.num_labels == 1 else nn.functional.mse_loss(logits, labels)
            elif self.config.problem_type == "single_label_classification":
                loss = nn.functional.cross_entropy(logits.view(-1, self.num_labels), labels.view(-1))
            else:
                loss = nn.functional.binary_cross_entropy_with_logits(logits, labels)
        if return_dict is False:
            return ((loss,) if loss is not None else ()) + (logits,) + ((hidden,) if want else ())
        return ImageClassifierOutput(loss=loss, logits=logits, hidden_states=hidden, attentions=None)


def encode_with(pretraining_model, pixel_values: torch.Tensor, bool_masked_pos: Optional[torch.Tensor] = None,
                output_hidden_states: Optional[bool] = None) -> BaseModelOutput:
    """``model.videomae(pixel_values)`` for a ``SmbVideoMAEForPreTraining`` (``src/run_inference.py:78-83``): its encoder over all
    tokens (or the visible ones), no decoder."""
    r = getattr(pretraining_model, "_encoder_runner", None)
    if r is None:
        r = _EncoderRunner(pretraining_model, pretraining_model.config, "videomae.", final_ln=False)
        object.__setattr__(pretraining_model, "_encoder_runner", r)
    if output_hidden_states:
        seq, hidden = r.encode(pixel_values, bool_masked_pos, True)
        return BaseModelOutput(last_hidden_state=seq, hidden_states=hidden)
    return BaseModelOutput(last_hidden_state=r.encode(pixel_values, bool_masked_pos))
