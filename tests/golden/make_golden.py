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


if __name__ == "__main__":
    make_masks()
    for n, s in GOLDEN_CONFIGS.items():
        make_mim(n, s)
