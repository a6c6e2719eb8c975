"""Generate the golden fixtures under tests/golden/.  Runs ONLY in the build container.

What it executes:

* G1 (masks): the reference ``MaskGenerator`` class, taken from the text of
  ``/root/reference/src/dataloader/mim.py`` (the module itself needs ``monai`` at import time, which is an
  ordinary ImportError here, so only that class — numpy + torch — is compiled from the file's AST and run).
* G2 (MIM step): the reference's model class for ``run_mim.py`` — ``transformers.VideoMAEForPreTraining``
  (``src/run_mim.py:13-21,345-357`` imports it from the installed ``transformers``; the vendored
  ``src/models/videomae/modeling_videomae.py`` is the same arithmetic but does not import under
  transformers 5.x) — built from a local ``VideoMAEConfig`` patched exactly as ``run_mim.py:302-330`` does
  (``num_channels=1, num_frames=depth, tubelet_size=patch_size``), fp32, eager attention, CPU.

Only DATA is written: inputs, weights, outputs, gradients, masks.  No reference source is copied.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")


def load_reference_mask_generator():
    path = os.path.join(REF, "src/dataloader/mim.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "MaskGenerator"]
    assert len(cls) == 1
    mod = ast.Module(body=cls, type_ignores=[])
    ns = {"np": np, "torch": torch}
    exec(compile(mod, path, "exec"), ns)
    return ns["MaskGenerator"]


MASK_CASES = [  # (input_size, depth, mask_patch_size, model_patch_size, ratio)
    (64, 32, 16, 16, 0.6),
    (64, 32, 32, 16, 0.6),
    (64, 32, 16, 8, 0.6),
    (224, 160, 32, 16, 0.5),
    (512, 320, 16, 16, 0.65),
    (512, 320, 32, 16, 0.65),
]


def make_masks():
    MG = load_reference_mask_generator()
    out = {}
    for (sz, dp, mps, ps, ratio) in MASK_CASES:
        gen = MG(input_size=sz, depth=dp, mask_patch_size=mps, model_patch_size=ps, mask_ratio=ratio)
        for seed in range(5):
            np.random.seed(seed)
            m1 = gen().numpy()
            m2 = gen().numpy()  # second draw from the same stream (the dataset draws one per sample)
            key = f"mask_{sz}_{dp}_{mps}_{ps}_{ratio}_seed{seed}"
            out[key + "_a"] = np.packbits(m1)
            out[key + "_b"] = np.packbits(m2)
            out[key + "_n"] = np.array([m1.size, int(m1.sum())], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **out)
    print("masks.npz:", len(out), "arrays")


def hf_state_from_pinned(p):
    """pinned-era names -> installed transformers (5.x) names: q_bias/v_bias -> query.bias/value.bias, key.bias=0."""
    out = {}
    for k, v in p.items():
        if k.endswith("attention.attention.q_bias"):
            out[k.replace("q_bias", "query.bias")] = v
            out[k.replace("q_bias", "key.bias")] = torch.zeros_like(v)
        elif k.endswith("attention.attention.v_bias"):
            out[k.replace("v_bias", "value.bias")] = v
        else:
            out[k] = v
    return out


def pinned_grads_from_hf(model):
    g = {}
    for k, v in model.named_parameters():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        if k.endswith("attention.attention.query.bias"):
            g[k.replace("query.bias", "q_bias")] = gr
        elif k.endswith("attention.attention.value.bias"):
            g[k.replace("value.bias", "v_bias")] = gr
        elif k.endswith("attention.attention.key.bias"):
            continue  # softmax-invariant; its gradient is ~0 and the pinned-era layout has no such parameter
        else:
            g[k] = gr
    return g


GOLDEN_CONFIGS = {
    # config 1 of BASELINE.json ("ViT-tiny" is build-chosen; the reference defines none): 64x64x32, patch 16
    "mim_tiny_p16": dict(cfg=dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=64,
                                  num_hidden_layers=2, num_attention_heads=1, intermediate_size=256,
                                  decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1,
                                  decoder_intermediate_size=128),
                         mask=(64, 32, 16, 16, 0.6), batch=2),
    # more tokens (N=256) at small weights: patch 8
    "mim_small_p8": dict(cfg=dict(image_size=64, num_frames=32, patch_size=8, tubelet_size=8, hidden_size=128,
                                  num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                                  decoder_hidden_size=64, decoder_num_hidden_layers=2, decoder_num_attention_heads=1,
                                  decoder_intermediate_size=256),
                         mask=(64, 32, 16, 8, 0.6), batch=2),
}


def make_mim(name, spec):
    from safetensors.torch import save_file
    from transformers import VideoMAEConfig, VideoMAEForPreTraining
    from oracle.mim_oracle import MIMConfig, init_params

    MG = load_reference_mask_generator()
    c = spec["cfg"]
    ocfg = MIMConfig(**c)
    hf_cfg = VideoMAEConfig(num_channels=1, **c)
    hf_cfg._attn_implementation = "eager"
    model = VideoMAEForPreTraining(hf_cfg).float().train()
    params = init_params(ocfg, seed=42, nonzero_small=True)
    missing, unexpected = model.load_state_dict(hf_state_from_pinned(params), strict=True), None
    B = spec["batch"]
    g = torch.Generator().manual_seed(1234)
    pix = torch.rand(B, ocfg.num_frames, 1, ocfg.image_size, ocfg.image_size, generator=g)
    sz, dp, mps, ps, ratio = spec["mask"]
    gen = MG(input_size=sz, depth=dp, mask_patch_size=mps, model_patch_size=ps, mask_ratio=ratio)
    np.random.seed(42)
    mask = torch.stack([gen() for _ in range(B)])
    out = model(pixel_values=pix, bool_masked_pos=mask)
    out.loss.backward()
    grads = pinned_grads_from_hf(model)
    blob = {"in.pixel_values": pix, "in.bool_masked_pos": mask.to(torch.uint8),
            "out.loss": out.loss.detach().reshape(1), "out.logits": out.logits.detach()}
    for k, v in params.items():
        blob["param." + k] = v.contiguous()
    for k, v in grads.items():
        blob["grad." + k] = v.detach().contiguous()
    meta = {"config": repr(c), "mask": repr(spec["mask"]), "generator": "transformers.VideoMAEForPreTraining fp32 eager CPU",
            "transformers": __import__("transformers").__version__, "torch": torch.__version__}
    save_file(blob, os.path.join(HERE, name + ".safetensors"), metadata=meta)
    print(name, "loss", float(out.loss), "logits", tuple(out.logits.shape), "n tensors", len(blob))


# ----------------------------------------------------------------------------------------------------------------------
# Encoder-only forwards: embedding export (VideoMAEModel, all tokens / visible tokens / with the final LayerNorm)
# and the classification head, from the installed transformers classes (the vendored file does not import under transformers 5.x).
# ----------------------------------------------------------------------------------------------------------------------
ENC_TINY = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                intermediate_size=256)


def _pinned_from_hf(sd):
    out = {}
    for k, v in sd.items():
        if k.endswith("attention.attention.query.bias"):
            out[k.replace("query.bias", "q_bias")] = v
        elif k.endswith("attention.attention.value.bias"):
            out[k.replace("value.bias", "v_bias")] = v
        elif k.endswith("attention.attention.key.bias"):
            continue
        else:
            out[k] = v
    return out


def _randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, p in model.named_parameters():
            if k.endswith("attention.attention.key.bias"):
                p.zero_()  # the pinned-era layout has no key bias (softmax-invariant anyway)
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif "norm" in k and k.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def make_encoder(name="encoder_tiny"):
    from safetensors.torch import save_file
    from transformers import VideoMAEConfig, VideoMAEForVideoClassification, VideoMAEModel

    MG = load_reference_mask_generator()
    blob = {}
    cfg = VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **ENC_TINY)
    cfg._attn_implementation = "eager"
    cls = VideoMAEForVideoClassification(cfg).float().eval()
    _randomise(cls, 7)
    g = torch.Generator().manual_seed(99)
    pix = torch.rand(2, 32, 1, 64, 64, generator=g)
    gen = MG(input_size=64, depth=32, mask_patch_size=16, model_patch_size=16, mask_ratio=0.6)
    np.random.seed(3)
    mask = torch.stack([gen() for _ in range(2)])
    with torch.no_grad():
        blob["cls.out.last_hidden_state"] = cls.videomae(pix).last_hidden_state
        blob["cls.out.visible_hidden_state"] = cls.videomae(pix, bool_masked_pos=mask).last_hidden_state
        blob["cls.out.logits"] = cls(pixel_values=pix).logits
    blob["in.pixel_values"], blob["in.bool_masked_pos"] = pix, mask.to(torch.uint8)
    for k, v in _pinned_from_hf(cls.state_dict()).items():
        blob["cls.param." + k] = v.contiguous()
    # fine-tuning step (src/run_classification.py): cross-entropy on labels, gradients of every parameter
    labels = torch.tensor([0, 2])
    out = cls(pixel_values=pix, labels=labels)
    out.loss.backward()
    blob["in.labels"], blob["cls.out.loss"] = labels, out.loss.detach().reshape(1)
    for k, v in pinned_grads_from_hf(cls).items():
        blob["cls.grad." + k] = v.detach().contiguous()
    cfg2 = VideoMAEConfig(num_channels=1, use_mean_pooling=False, **ENC_TINY)
    cfg2._attn_implementation = "eager"
    enc = VideoMAEModel(cfg2).float().eval()
    _randomise(enc, 8)
    with torch.no_grad():
        blob["enc.out.last_hidden_state"] = enc(pix).last_hidden_state
    for k, v in _pinned_from_hf(enc.state_dict()).items():
        blob["enc.param." + k] = v.contiguous()
    meta = {"config": repr(ENC_TINY), "generator": "transformers.VideoMAEForVideoClassification / VideoMAEModel fp32 eager CPU",
            "transformers": __import__("transformers").__version__, "torch": torch.__version__}
    save_file({k: v.contiguous() for k, v in blob.items()}, os.path.join(HERE, name + ".safetensors"), metadata=meta)
    print(name, "n tensors", len(blob), "logits", blob["cls.out.logits"])


# ----------------------------------------------------------------------------------------------------------------------
# V-JEPA (src/run_vjepa.py): the in-repo model class src/models/vjepa/modeling_vjepa.py imports as shipped.
# ----------------------------------------------------------------------------------------------------------------------

VJEPA_TINY = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=2,
                  num_attention_heads=2, mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2,
                  pred_num_mask_tokens=10, pred_mlp_ratio=4.0)


def load_reference_vjepa_mask_generator():
    """VJEPAMaskGenerator from the text of src/dataloader/transforms.py (module import needs monai); its MONAI ``Transform`` base only
    contributes ``__init__``/call plumbing, so the class is compiled with ``object`` as base."""
    import math
    path = os.path.join(REF, "src/dataloader/transforms.py")
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VJEPAMaskGenerator"]
    assert len(cls) == 1
    cls[0].bases = []
    mod = ast.Module(body=cls, type_ignores=[])
    ns = {"np": np, "torch": torch, "math": math}
    exec(compile(mod, path, "exec"), ns)
    return ns["VJEPAMaskGenerator"]


def make_vjepa_masks():
    G = load_reference_vjepa_mask_generator()
    out = {}
    cases = [((8, 32, 32), (2, 8, 8), 1), ((8, 32, 32), (2, 8, 8), 3), ((160, 224, 224), (16, 16, 16), 3), ((320, 512, 512), (16, 16, 16), 3)]
    for ci, (inp, ps, nb) in enumerate(cases):
        gen = G(input_size=inp, patch_size=ps, pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=nb)
        for seed in range(4):
            torch.manual_seed(seed)
            for draw in range(2):
                d = gen({})
                key = f"vmask_{ci}_seed{seed}_{draw}"
                out[key + "_ctx"] = d["context_mask"].reshape(-1).numpy().astype(np.int64)
                out[key + "_tgt"] = d["target_mask"].reshape(-1).numpy().astype(np.int64)
        out[f"vmask_{ci}_cfg"] = np.array(list(inp) + list(ps) + [nb], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vjepa_masks.npz"), **out)
    print("vjepa_masks.npz:", len(out), "arrays")


def make_vjepa(name="vjepa_tiny"):
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(REF, "src"))
    from models.vjepa.configuration_vjepa import VJEPA2Config
    from models.vjepa.modeling_vjepa import VJEPA2Model
    from oracle.vjepa_oracle import VJEPAConfig, init_params

    ocfg = VJEPAConfig(**VJEPA_TINY)
    rcfg = VJEPA2Config(**VJEPA_TINY)
    rcfg._attn_implementation = "eager"
    online, target = VJEPA2Model(rcfg).float().train(), VJEPA2Model(rcfg).float().eval()
    p_on, p_tg = init_params(ocfg, seed=3), init_params(ocfg, seed=4)

    def load(model, p):
        sd = model.state_dict()
        for k in sd:
            if k in p:
                sd[k] = p[k].clone()
        model.load_state_dict(sd)
    load(online, p_on)
    load(target, p_tg)
    B = 2
    pix = torch.rand(B, ocfg.frames_per_clip, 1, ocfg.crop_size, ocfg.crop_size, generator=torch.Generator().manual_seed(7))
    G = load_reference_vjepa_mask_generator()
    gen = G(input_size=(ocfg.frames_per_clip, ocfg.crop_size, ocfg.crop_size), patch_size=(ocfg.tubelet_size, ocfg.patch_size, ocfg.patch_size),
            pred_mask_scale=(0.2, 0.5), aspect_ratio=(0.3, 3.0), num_blocks=2)
    torch.manual_seed(11)
    d = gen({})
    cm = torch.stack([d["context_mask"].reshape(-1)] * B)  # run_vjepa.py:144-160: one example's masks for the whole batch
    tm = torch.stack([d["target_mask"].reshape(-1)] * B)
    out = online(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm], skip_predictor=False)
    pred = out.predictor_output.last_hidden_state
    with torch.no_grad():
        tgt = target(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm], skip_predictor=True).target_hidden_state
    loss = torch.nn.L1Loss()(pred, tgt)
    loss.backward()
    blob = {"in.pixel_values_videos": pix, "in.context_mask": cm, "in.target_mask": tm, "out.loss": loss.detach().reshape(1),
            "out.last_hidden_state": out.last_hidden_state.detach(), "out.predictor": pred.detach(), "out.target_hidden_state": tgt}
    names = set(p_on)
    for k, v in p_on.items():
        blob["param." + k] = v.contiguous()
    for k, v in p_tg.items():
        blob["tparam." + k] = v.contiguous()
    for k, v in online.named_parameters():
        if k in names:
            blob["grad." + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).detach().contiguous()
    save_file(blob, os.path.join(HERE, name + ".safetensors"), metadata={"config": repr(VJEPA_TINY), "generator": "reference VJEPA2Model fp32 eager CPU"})
    print(name, "loss", float(loss), "pred", tuple(pred.shape), "ctx/tgt", cm.shape[1], tm.shape[1], "tensors", len(blob))


VJEPA_MULTI = dict(patch_size=8, crop_size=32, frames_per_clip=4, tubelet_size=2, in_chans=1, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                   mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)


def make_vjepa_multimask(name="vjepa_multimask"):
    """Lists of TWO context and TWO target masks (the case of the reference's tests/test_vjepa_model.py:96-125, same sizes: batch 2, 32 tokens,
    20 context / 12 target ids drawn as independent random permutations): all four outputs of the reference VJEPA2Model and the gradients of a
    fixed linear functional of the predictor output."""
    from safetensors.torch import save_file
    sys.path.insert(0, os.path.join(REF, "src"))
    from models.vjepa.configuration_vjepa import VJEPA2Config
    from models.vjepa.modeling_vjepa import VJEPA2Model
    from oracle.vjepa_oracle import VJEPAConfig, init_params
    ocfg = VJEPAConfig(**VJEPA_MULTI)
    rcfg = VJEPA2Config(**VJEPA_MULTI)
    rcfg._attn_implementation = "eager"
    model = VJEPA2Model(rcfg).float().train()
    p = init_params(ocfg, seed=5, nonzero_small=True)
    sd = model.state_dict()
    for k in sd:
        if k in p:
            sd[k] = p[k].clone()
    model.load_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    pix = torch.rand(2, 4, 1, 32, 32, generator=g)
    cms = [torch.randperm(32, generator=g)[:20].unsqueeze(0).repeat(2, 1) for _ in range(2)]
    tms = [torch.randperm(32, generator=g)[20:].unsqueeze(0).repeat(2, 1) for _ in range(2)]
    w = torch.randn(4, 12, 64, generator=g)
    out = model(pixel_values_videos=pix, context_mask=cms, target_mask=tms)
    (out.predictor_output.last_hidden_state * w).sum().backward()
    blob = {"in.pixel_values_videos": pix, "in.context_mask": torch.stack(cms), "in.target_mask": torch.stack(tms), "in.w": w,
            "out.last_hidden_state": out.last_hidden_state.detach(), "out.masked_hidden_state": out.masked_hidden_state.detach(),
            "out.target_hidden_state": out.target_hidden_state.detach(), "out.predictor": out.predictor_output.last_hidden_state.detach().contiguous()}
    for k, v in p.items():
        blob["param." + k] = v.contiguous()
    for k, v in model.named_parameters():
        if k in p:
            blob["grad." + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).detach().contiguous()
    save_file(blob, os.path.join(HERE, name + ".safetensors"), metadata={"config": repr(VJEPA_MULTI), "generator": "reference VJEPA2Model fp32 eager CPU, 2 mask pairs"})
    print(name, {k: tuple(v.shape) for k, v in blob.items() if k.startswith("out.")})


def make_decay_names(name="decay_names"):
    """The decay / no-decay split HF ``Trainer.get_decay_parameter_names`` computes for the reference model classes (``run_mim.py`` /
    ``run_vjepa.py`` hand those to ``Trainer``; ``scripts/training/run_vjepa.sh:23`` sets --weight_decay 0.01): parameter NAMES only, as JSON."""
    import json
    from transformers import Trainer, VideoMAEConfig, VideoMAEForPreTraining
    sys.path.insert(0, os.path.join(REF, "src"))
    from models.vjepa.configuration_vjepa import VJEPA2Config
    from models.vjepa.modeling_vjepa import VJEPA2Model
    out = {}
    mim = VideoMAEForPreTraining(VideoMAEConfig(num_channels=1, **GOLDEN_CONFIGS["mim_tiny_p16"]["cfg"]))
    dec = set(Trainer.get_decay_parameter_names(None, mim))
    pin = lambda k: k.replace("attention.attention.query.bias", "attention.attention.q_bias").replace("attention.attention.value.bias", "attention.attention.v_bias")
    names = [k for k, _ in mim.named_parameters() if not k.endswith("attention.attention.key.bias")]
    out["mim"] = {"decay": sorted(pin(k) for k in names if k in dec), "no_decay": sorted(pin(k) for k in names if k not in dec)}
    vj = VJEPA2Model(VJEPA2Config(**VJEPA_TINY))
    dec = set(Trainer.get_decay_parameter_names(None, vj))
    names = [k for k, _ in vj.named_parameters()]
    out["vjepa"] = {"decay": sorted(k for k in names if k in dec), "no_decay": sorted(k for k in names if k not in dec)}
    out["_generator"] = "transformers %s Trainer.get_decay_parameter_names on transformers.VideoMAEForPreTraining / reference VJEPA2Model" % __import__("transformers").__version__
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(name, {k: (len(v["decay"]), len(v["no_decay"])) for k, v in out.items() if not k.startswith("_")})


def make_hidden_states(name="hidden_states_tiny"):
    """output_hidden_states=True of the installed transformers classes on the EXISTING fixtures' weights and inputs (encoder_tiny, mim_tiny_p16): the
    classes are rebuilt from the stored parameters, must reproduce the stored outputs bit for bit, and only the hidden-state tuples are written."""
    from safetensors.torch import load_file, save_file
    from transformers import VideoMAEConfig, VideoMAEForPreTraining, VideoMAEForVideoClassification, VideoMAEModel

    out = {}
    enc = load_file(os.path.join(HERE, "encoder_tiny.safetensors"))
    pix, mask = enc["in.pixel_values"], enc["in.bool_masked_pos"].bool()
    cfg = VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **ENC_TINY)
    cfg._attn_implementation = "eager"
    cls = VideoMAEForVideoClassification(cfg).float().eval()
    cls.load_state_dict(hf_state_from_pinned({k[len("cls.param."):]: v for k, v in enc.items() if k.startswith("cls.param.")}), strict=True)
    cfg2 = VideoMAEConfig(num_channels=1, use_mean_pooling=False, **ENC_TINY)
    cfg2._attn_implementation = "eager"
    mdl = VideoMAEModel(cfg2).float().eval()
    mdl.load_state_dict(hf_state_from_pinned({k[len("enc.param."):]: v for k, v in enc.items() if k.startswith("enc.param.")}), strict=True)
    with torch.no_grad():
        a = cls.videomae(pix, output_hidden_states=True)
        assert torch.equal(a.last_hidden_state, enc["cls.out.last_hidden_state"])
        b = cls.videomae(pix, bool_masked_pos=mask, output_hidden_states=True)
        assert torch.equal(b.last_hidden_state, enc["cls.out.visible_hidden_state"])
        c = mdl(pix, output_hidden_states=True)
        assert torch.equal(c.last_hidden_state, enc["enc.out.last_hidden_state"])
        full = cls(pixel_values=pix, output_hidden_states=True)
        assert torch.equal(full.logits, enc["cls.out.logits"]) and len(full.hidden_states) == len(a.hidden_states)
    for tag, o in (("cls.all", a), ("cls.visible", b), ("enc.all", c)):
        assert len(o.hidden_states) == ENC_TINY["num_hidden_layers"] + 1
        for l, h in enumerate(o.hidden_states):
            out[f"{tag}.{l}"] = h.contiguous()
    mim = load_file(os.path.join(HERE, "mim_tiny_p16.safetensors"))
    mcfg = VideoMAEConfig(num_channels=1, **GOLDEN_CONFIGS["mim_tiny_p16"]["cfg"])
    mcfg._attn_implementation = "eager"
    m = VideoMAEForPreTraining(mcfg).float().eval()
    m.load_state_dict(hf_state_from_pinned({k[len("param."):]: v for k, v in mim.items() if k.startswith("param.")}), strict=True)
    with torch.no_grad():
        o = m(pixel_values=mim["in.pixel_values"], bool_masked_pos=mim["in.bool_masked_pos"].bool(), output_hidden_states=True)
    assert torch.equal(o.logits, mim["out.logits"]) and float(o.loss) == float(mim["out.loss"])
    assert len(o.hidden_states) == mcfg.num_hidden_layers + 1
    for l, h in enumerate(o.hidden_states):
        out[f"mim.{l}"] = h.contiguous()
    meta = {"generator": "transformers VideoMAEModel / VideoMAEForVideoClassification / VideoMAEForPreTraining fp32 eager CPU, output_hidden_states=True, on the "
                         "parameters and inputs of encoder_tiny.safetensors / mim_tiny_p16.safetensors", "transformers": __import__("transformers").__version__, "torch": torch.__version__}
    save_file(out, os.path.join(HERE, name + ".safetensors"), metadata=meta)
    print(name, {k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["mim", "vjepa", "encoder", "hidden"]
    if "mim" in which:
        make_masks()
        for n, s in GOLDEN_CONFIGS.items():
            make_mim(n, s)
    if "vjepa" in which:
        make_vjepa_masks()
        make_vjepa()
    if "vjepa" in which or "vjepa_multimask" in which:
        make_vjepa_multimask()
    if "encoder" in which:
        make_encoder()
    if "hidden" in which:
        make_hidden_states()
    if "decay" in which:
        make_decay_names()
