"""Seeded random sweep of MIM step configurations against the CPU oracle (fp32): widths that do and do not meet the grouped weight-gradient launch's
conditions (multiples of 128 or not), head widths 16 / 32 / 48 / 64 (native and zero-padded builds), batch 1-3, mask ratios 0.3-0.8, mask patches of one or two
model patches, recompute on / off, the attention residual on / off.  The fixed-shape tests of test_mim_step_gpu.py pin the shapes the step was tuned on; this
one walks the dispatch conditions between them (block.hip: deferred vs inline weight gradients, padded heads, ranged last block, two activation sets).
Tolerances: those of test_mim_step_gpu.py (loss 5e-3, logits rms 1e-2 / max 3e-2, gradients 6e-2 per tensor, cosine >= 0.998)."""
import random

import numpy as np
import pytest
import torch

from oracle.mim_oracle import MIMConfig, init_params, mim_step
from test_mim_step_gpu import check_against, run_step

pytestmark = pytest.mark.gpu


def draw(seed):
    r = random.Random(seed)
    heads_for = {64: [1, 2, 4], 96: [2, 3, 6], 128: [2, 4, 8], 192: [3, 4, 6], 256: [4, 8], 384: [6, 12]}
    hid = r.choice([64, 96, 128, 192, 256, 384])
    dec = r.choice([64, 96, 128, 192, 256])
    mp = r.choice([16, 16, 32])
    image = r.choice([64, 96, 128] if mp == 16 else [64, 128])
    frames = r.choice([32, 48, 64] if mp == 16 else [32, 64])
    c = dict(image_size=image, num_frames=frames, patch_size=16, tubelet_size=16, hidden_size=hid, num_hidden_layers=r.randint(1, 3),
             num_attention_heads=r.choice(heads_for[hid]), intermediate_size=r.choice([128, 192, 256, 320, 512]), decoder_hidden_size=dec,
             decoder_num_hidden_layers=r.randint(1, 3), decoder_num_attention_heads=r.choice(heads_for[dec]),
             decoder_intermediate_size=r.choice([128, 192, 256, 384]))
    return c, dict(B=r.randint(1, 3), ratio=r.uniform(0.3, 0.8), mask_patch=mp, recompute=r.random() < 0.4, resid=r.random() < 0.5, seed=seed)


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_step_configuration_matches_oracle(seed):
    from transformers import VideoMAEConfig
    from smb_vision_amd.masking import MaskGenerator
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    from smb_vision_amd import ops
    c, o = draw(seed)
    cfg = MIMConfig(**c)
    params = init_params(cfg, seed=100 + seed, nonzero_small=True)
    np.random.seed(seed)
    gen = MaskGenerator(c["image_size"], c["num_frames"], o["mask_patch"], 16, o["ratio"])
    mask = torch.stack([gen() for _ in range(o["B"])])
    pix = torch.rand(o["B"], c["num_frames"], 1, c["image_size"], c["image_size"], generator=torch.Generator().manual_seed(seed))
    ref_loss, ref_logits, ref_grads = mim_step(params, cfg, pix, mask)
    extra = dict(smbv_attention_residual=o["resid"])
    if o["recompute"]:
        extra["smbv_recompute"] = "always"
    model = SmbVideoMAEForPreTraining(VideoMAEConfig(num_channels=1, **c, **extra))
    model.load_state_dict(params, strict=True)
    model = model.cuda().train()
    if o["recompute"]:
        model.gradient_checkpointing_enable()
    loss, logits, grads = run_step(model, pix, mask)
    worst = check_against(loss, logits, grads, ref_loss, ref_logits, ref_grads)
    loss2, logits2, grads2 = run_step(model, pix, mask)   # and the configuration's step is run-to-run identical
    assert torch.equal(loss, loss2) and torch.equal(logits, logits2) and all(torch.equal(grads[k], grads2[k]) for k in grads)
    assert ops.device_status() == 0
    print(f"seed {seed}: {c} {o}: loss {float(loss):.6f} oracle {float(ref_loss):.6f} worst grad {worst}")


def draw_vjepa(seed):
    r = random.Random(1000 + seed)
    heads_for = {64: [1, 2, 4], 96: [2, 3, 6], 128: [2, 4, 8], 192: [3, 4, 6]}
    hid, pred = r.choice([64, 96, 128, 192]), r.choice([64, 96, 128])
    c = dict(patch_size=8, crop_size=r.choice([32, 48, 64]), frames_per_clip=r.choice([8, 12, 16]), tubelet_size=2, in_chans=1, hidden_size=hid,
             num_hidden_layers=r.randint(1, 3), num_attention_heads=r.choice(heads_for[hid]), mlp_ratio=r.choice([2.0, 4.0]), pred_hidden_size=pred,
             pred_num_hidden_layers=r.randint(1, 3), pred_num_attention_heads=r.choice(heads_for[pred]), pred_num_mask_tokens=10,
             pred_mlp_ratio=r.choice([2.0, 4.0]))
    return c, dict(B=r.randint(1, 3), ctx=r.uniform(0.2, 0.6), n_masks=r.choice([1, 1, 2]), seed=seed)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_vjepa_configuration_matches_oracle(seed):
    """The same walk for the V-JEPA step (online encoder + predictor against the EMA target's L1 loss): head widths 16 .. 64 on both stacks, one or two
    (context, target) mask pairs, batch 1-3.  Tolerances of tests/test_vjepa_gpu.py (key biases: cancellation-dominated under RoPE, 2e-1)."""
    import oracle.vjepa_oracle as vo
    from transformers import VJEPA2Config
    from smb_vision_amd import ops
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model, make_target_encoder, vjepa_loss
    c, o = draw_vjepa(seed)
    cfg = vo.VJEPAConfig(**c)
    params = vo.init_params(cfg, seed=200 + seed, nonzero_small=True)
    n = cfg.num_patches
    g = torch.Generator().manual_seed(seed)
    cms, tms = [], []
    for _ in range(o["n_masks"]):   # every volume of the batch shares the pair's token counts (the collate's contract), not its tokens
        k = max(1, int(n * o["ctx"]))
        perms = [torch.randperm(n, generator=g) for _ in range(o["B"])]
        cms.append(torch.stack([p[:k].sort().values for p in perms]))
        tms.append(torch.stack([p[k:].sort().values for p in perms]))
    pix = torch.rand(o["B"], c["frames_per_clip"], 1, c["crop_size"], c["crop_size"], generator=g)
    model = SmbVJEPA2Model(VJEPA2Config(**c))
    model.load_state_dict(params, strict=True)
    model = model.cuda().train()
    target = make_target_encoder(model)
    inputs = {"pixel_values_videos": pix.cuda(), "context_mask": [m.cuda() for m in cms], "target_mask": [m.cuda() for m in tms]}
    loss, out = vjepa_loss(model, target, None, inputs)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_pred, _, ref_grads = vo.vjepa_step(params, params, cfg, pix, cms, tms)
    assert abs(float(loss) - float(ref_loss)) < 1e-2 * float(ref_loss), (float(loss), float(ref_loss))
    worst = ("", 0.0)
    for k, p_ in model.named_parameters():
        ref = ref_grads[k]
        nrm = float(ref.norm())
        if nrm < 1e-10:
            assert float(p_.grad.norm()) < 1e-6, k
            continue
        e = float((p_.grad.cpu() - ref).norm()) / nrm
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < (2e-1 if k.endswith("attention.key.bias") else 8e-2), (k, e, nrm)
    assert ops.device_status() == 0
    print(f"vjepa seed {seed}: {c} {o}: loss {float(loss):.6f} oracle {float(ref_loss):.6f} worst grad {worst}")
