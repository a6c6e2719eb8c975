"""ctypes binding of libsmbv.so (include/smbv.h).  There is NO fallback: if the library is missing or a call fails
this raises — the product path never routes around the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsmbv.so")
MAX_LAYERS = 48
ABI_MAJOR, ABI_MINOR_NEEDED = 3, 6  # include/smbv.h SMBV_ABI_MAJOR; the newest entry point these bindings call

c_f = C.c_void_p  # all device pointers travel as void*


class GemmEpi(C.Structure):
    _fields_ = [
        ("alpha", C.c_float), ("bias", c_f),
        ("rowadd", c_f), ("rowidx", c_f), ("ldr", C.c_int),
        ("dgelu_pre", c_f), ("ld_pre", C.c_int),
        ("act_gelu", C.c_int), ("aux_pre", c_f), ("ld_aux", C.c_int),
        ("out_bf16", c_f), ("ld_ob", C.c_int),
        ("out_f32", c_f), ("ld_of", C.c_int),
        ("grp_in", C.c_int), ("grp_out", C.c_int),
    ]


class TnJob(C.Structure):
    _fields_ = [("X", c_f), ("Y", c_f), ("out", c_f), ("sum_dst", c_f * 3), ("ldx", C.c_int), ("ldy", C.c_int), ("M", C.c_int), ("N1", C.c_int),
                ("N2", C.c_int), ("seg", C.c_int), ("reserved", C.c_int)]


class Rope3dCfg(C.Structure):
    _fields_ = [("table", c_f), ("pos", c_f), ("S", C.c_int), ("grid", C.c_int), ("Dh", C.c_int), ("head_stride", C.c_int), ("table_rows", C.c_int)]


class _Cfg(C.Structure):
    """cfg structs open with ``struct_bytes`` (ABI 2000): filled in here so that a binding / library layout mismatch is refused by the library."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_bytes = C.sizeof(type(self))


class MimCfg(_Cfg):
    _fields_ = [(n, C.c_int) for n in
                ("struct_bytes", "B", "T", "H", "W", "ps", "ts", "N", "Nvis", "Nmask", "d", "heads", "ff", "layers", "dd", "dheads", "dff", "dlayers")] + \
               [("ln_eps", C.c_float), ("dec_norm_eps", C.c_float), ("recompute", C.c_int), ("attn_fp8", C.c_int), ("attn_o_resid", C.c_int)]


LAYER_FIELDS = ("ln1_w", "ln1_b", "q_w", "k_w", "v_w", "q_b", "v_b", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class LayerPtrs(C.Structure):
    _fields_ = [(n, c_f) for n in LAYER_FIELDS]


class MimTensors(C.Structure):
    _fields_ = [("pe_w", c_f), ("pe_b", c_f), ("enc", LayerPtrs * MAX_LAYERS), ("e2d_w", c_f), ("mask_token", c_f),
                ("dec", LayerPtrs * MAX_LAYERS), ("dnorm_w", c_f), ("dnorm_b", c_f), ("head_w", c_f), ("head_b", c_f)]


class MimIO(C.Structure):
    _fields_ = [("pixel_values", c_f), ("vis_idx", c_f), ("mask_idx", c_f), ("pos_enc", c_f), ("pos_dec", c_f),
                ("loss", c_f), ("logits", c_f), ("grad_scale", c_f)]


class EncoderCfg(_Cfg):
    _fields_ = [(n, C.c_int) for n in ("struct_bytes", "B", "T", "H", "W", "ps", "ts", "N", "Nsel", "d", "heads", "ff", "layers")] + \
               [("ln_eps", C.c_float), ("final_ln", C.c_int), ("final_ln_eps", C.c_float), ("train", C.c_int), ("attn_o_resid", C.c_int)]


class EncoderTensors(C.Structure):
    _fields_ = [("pe_w", c_f), ("pe_b", c_f), ("enc", LayerPtrs * MAX_LAYERS), ("lnf_w", c_f), ("lnf_b", c_f)]


class EncoderIO(C.Structure):
    _fields_ = [("pixel_values", c_f), ("sel_idx", c_f), ("pos_enc", c_f), ("last_hidden_state", c_f)]


class AdamChunk(C.Structure):
    _fields_ = [("p", c_f), ("g", c_f), ("m", c_f), ("v", c_f), ("n", C.c_int32), ("group", C.c_int32)]


class AdamHyper(C.Structure):
    _fields_ = [("lr", C.c_float * 8), ("weight_decay", C.c_float * 8), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias_corr1", C.c_float), ("bias_corr2", C.c_float), ("max_grad_norm", C.c_float)]


class VjepaCfg(_Cfg):
    _fields_ = [(n, C.c_int) for n in ("struct_bytes", "B", "T", "H", "W", "ps", "ts", "N", "Nctx", "Ntgt", "d", "heads", "ff", "layers", "pd", "pheads", "pff", "players")] + \
               [("ln_eps", C.c_float), ("n_masks", C.c_int), ("enc_needed", C.c_int), ("pred_tgt_first", C.c_int), ("attn_o_resid", C.c_int)]


BLOCK_FIELDS = ("ln1_w", "ln1_b", "q_w", "k_w", "v_w", "q_b", "k_b", "v_b", "proj_w", "proj_b", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class BlockPtrs(C.Structure):
    _fields_ = [(n, c_f) for n in BLOCK_FIELDS]


class VjepaTensors(C.Structure):
    _fields_ = [("pe_w", c_f), ("pe_b", c_f), ("enc", BlockPtrs * MAX_LAYERS), ("enc_ln_w", c_f), ("enc_ln_b", c_f), ("pemb_w", c_f), ("pemb_b", c_f),
                ("mask_token", c_f), ("pred", BlockPtrs * MAX_LAYERS), ("pred_ln_w", c_f), ("pred_ln_b", c_f), ("pproj_w", c_f), ("pproj_b", c_f)]


class VjepaIO(C.Structure):
    _fields_ = [(n, c_f) for n in ("pixel_values", "all_idx", "ctx_idx", "tgt_idx", "pred_pos", "ctx_slot", "tgt_slot", "rope_enc", "rope_pred", "enc_out",
                                   "pred_out", "d_pred", "d_enc", "enc_pos")]


_SIGS = {
    "smbv_version": (C.c_int, []),
    "smbv_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "smbv_debug_set_variant": (None, [C.c_int]),
    "smbv_device_status": (C.c_int, [C.c_int]),
    "smbv_debug_pair_state": (C.c_int, [C.c_int, c_f]),
    "smbv_gemm_nt": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GemmEpi), c_f]),
    "smbv_gemm_tn": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_size_t, c_f]),
    "smbv_gemm_tn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "smbv_gemm_nt_splitk": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GemmEpi), c_f, C.c_size_t, c_f]),
    "smbv_gemm_nt_splitk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "smbv_gemm_nt_rope": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GemmEpi), C.POINTER(Rope3dCfg), C.c_int, c_f]),
    "smbv_gemm_tn_grouped": (C.c_int, [C.POINTER(TnJob), C.c_int, c_f]),
    "smbv_gemm_tn_grouped_split_workspace_bytes": (C.c_size_t, [C.POINTER(TnJob), C.c_int]),
    "smbv_gemm_tn_grouped_split": (C.c_int, [C.POINTER(TnJob), C.c_int, c_f, C.c_size_t, c_f]),
    "smbv_gemm_tn_bias": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, c_f, C.c_size_t, c_f]),
    "smbv_layernorm_fwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_float, C.c_long, C.c_long, C.c_long, c_f]),
    "smbv_layernorm_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, c_f, C.c_size_t, c_f]),
    "smbv_layernorm_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "smbv_attention_fwd": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, c_f]),
    "smbv_attention_fp8_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "smbv_attention_fwd_fp8": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, c_f, C.c_size_t, c_f]),
    "smbv_attention_bwd": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_attention_bwd_split_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "smbv_attention_bwd_split": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_int, C.c_int, c_f, C.c_size_t, c_f, C.c_size_t, c_f]),
    "smbv_attention_bwd_rope": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                          C.c_int, C.c_int, c_f, C.c_size_t, c_f, C.c_size_t, C.POINTER(Rope3dCfg), c_f]),
    "smbv_attention_bwd_fp8_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "smbv_attention_bwd_fp8": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                         C.c_int, C.c_int, c_f, C.c_size_t, c_f]),
    "smbv_prepare_volume": (C.c_int, [c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, c_f]),
    "smbv_gather_bricks": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_brick_stats": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_mse_fwd": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_mse_bwd": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_cast_f32_bf16": (C.c_int, [c_f, c_f, C.c_size_t, c_f]),
    "smbv_transpose_cast_f32_bf16": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "smbv_colsum_bf16": (C.c_int, [c_f, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_size_t, c_f]),
    "smbv_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "smbv_grad_sqnorm": (C.c_int, [c_f, C.c_int, c_f, c_f, c_f]),
    "smbv_adamw_step": (C.c_int, [c_f, C.c_int, C.POINTER(AdamHyper), c_f, c_f]),
    "smbv_rope3d": (C.c_int, [c_f, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f, C.c_int, c_f]),
    "smbv_gather_rows_idx_f32": (C.c_int, [c_f, c_f, c_f, C.c_long, C.c_int, C.c_long, C.c_int, c_f]),
    "smbv_scatter_rows_idx_f32": (C.c_int, [c_f, c_f, c_f, C.c_long, C.c_int, C.c_long, C.c_int, c_f]),
    "smbv_fill_rows_idx_f32": (C.c_int, [c_f, c_f, c_f, C.c_long, C.c_int, C.c_long, C.c_int, c_f]),
    "smbv_l1_fwd": (C.c_int, [c_f, c_f, C.c_size_t, c_f, c_f, c_f]),
    "smbv_l1_bwd": (C.c_int, [c_f, c_f, C.c_size_t, c_f, c_f, c_f]),
    "smbv_ema_update": (C.c_int, [c_f, C.c_int, C.c_float, c_f]),
    "smbv_vjepa_workspace_bytes": (C.c_size_t, [C.POINTER(VjepaCfg)]),
    "smbv_vjepa_forward": (C.c_int, [C.POINTER(VjepaCfg), C.POINTER(VjepaTensors), C.POINTER(VjepaIO), c_f, C.c_size_t, c_f]),
    "smbv_vjepa_backward": (C.c_int, [C.POINTER(VjepaCfg), C.POINTER(VjepaTensors), C.POINTER(VjepaTensors), C.POINTER(VjepaIO), c_f, C.c_size_t,
                                      C.POINTER(C.c_void_p), C.c_int, c_f]),
    "smbv_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderCfg)]),
    "smbv_encoder_forward": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(EncoderTensors), C.POINTER(EncoderIO), c_f, C.c_size_t, c_f]),
    "smbv_encoder_backward": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(EncoderTensors), C.POINTER(EncoderTensors), C.POINTER(EncoderIO), c_f, c_f, C.c_size_t, c_f]),
    "smbv_encoder_hidden_states": (C.c_int, [C.POINTER(EncoderCfg), c_f, C.c_size_t, C.POINTER(C.c_void_p), C.c_int]),
    "smbv_mim_hidden_states": (C.c_int, [C.POINTER(MimCfg), c_f, C.c_size_t, C.POINTER(C.c_void_p), C.c_int]),
    "smbv_mim_workspace_bytes": (C.c_size_t, [C.POINTER(MimCfg)]),
    "smbv_mim_forward": (C.c_int, [C.POINTER(MimCfg), C.POINTER(MimTensors), C.POINTER(MimIO), c_f, C.c_size_t, c_f]),
    "smbv_mim_backward": (C.c_int, [C.POINTER(MimCfg), C.POINTER(MimTensors), C.POINTER(MimTensors), C.POINTER(MimIO), c_f, C.c_size_t,
                                    C.POINTER(C.c_void_p), C.c_int, c_f]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class SmbvError(RuntimeError):
    pass


def load():
    """Load libsmbv.so or raise.  (Build it with ``python -m smb_vision_amd.build``.)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmbvError(f"{LIB_PATH} not found: the smb_vision_amd HIP library is required (python -m smb_vision_amd.build); "
                        "there is no CPU / PyTorch fallback for the hot path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if an export declared in include/smbv.h is missing
        fn.restype = res
        fn.argtypes = args
    ver = lib.smbv_version()
    if ver // 1000 != ABI_MAJOR or ver % 1000 < ABI_MINOR_NEEDED:
        raise SmbvError(f"{LIB_PATH} reports ABI {ver}; these bindings need major {ABI_MAJOR}, minor >= {ABI_MINOR_NEEDED} "
                        "(rebuild with python -m smb_vision_amd.build)")
    _lib = lib
    return lib


def detach_aliased_grads(params, arena, offsets=None, owner=None):
    """Gradient accumulation support.  A step's gradients are written into ONE persistent flat arena and handed to autograd as views of
    it; when ``p.grad`` is None autograd keeps such a view as ``p.grad`` (no copy).  If a parameter still holds that alias when the
    next backward starts (``gradient_accumulation_steps > 1`` as scripts/training/run_vjepa.sh:28 sets, or ``zero_grad(set_to_none=
    False)``), the arena is about to be overwritten underneath it: the parameter needs its own storage first, so that autograd's
    ``p.grad += view`` afterwards is the sum the reference computes.  Costs nothing when every ``p.grad`` is None.

    With ``offsets`` / ``owner`` (the model) the private storage is ONE persistent accumulation arena kept on the model (``owner._gacc``): one
    device copy of the whole arena and ``p.grad`` re-pointed to views of it at the same offsets.  Their addresses never change from cycle to cycle, so
    SmbvAdamW's device-side chunk table is built once (per-parameter clones gave new pointers, and a table rebuild + upload, every step)."""
    lo = arena.data_ptr()
    hi = lo + arena.numel() * arena.element_size()
    aliased = [i for i, p in enumerate(params) if p.grad is not None and lo <= p.grad.data_ptr() < hi]
    if not aliased:
        return
    if offsets is None or owner is None:
        for i in aliased:
            params[i].grad = params[i].grad.clone()
        return
    acc = getattr(owner, "_gacc", None)
    if acc is None or acc.shape != arena.shape or acc.device != arena.device:
        acc = torch.empty_like(arena)
        owner._gacc = acc
    whole = len(aliased) == len(params)  # the usual case: every parameter still holds this step's view -> one copy of the arena
    if whole:
        acc.copy_(arena)
    for i in aliased:
        p = params[i]
        view = acc[offsets[i]:offsets[i] + p.numel()].view(p.shape)
        if not whole:  # (some parameters already accumulate in `acc`: their slices must stay)
            view.copy_(p.grad)
        p.grad = view


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().smbv_last_error(buf, 512)
    return buf.value.decode()


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = last_error()
    if rc in (-1,):
        raise ValueError(f"{what}: {msg}")
    raise SmbvError(f"{what} failed (rc={rc}): {msg}")
