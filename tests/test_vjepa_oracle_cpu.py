"""CPU: pin the V-JEPA oracle against golden vectors produced by the reference's own classes (VJEPA2Model, VJEPAMaskGenerator)."""
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle.vjepa_oracle import (VJEPAConfig, VJEPAMaskGeneratorOracle, ema_update, init_params, model_forward, param_shapes, rope3d, rotate,
                                 vjepa_step)

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    blob = {}
    with safe_open(os.path.join(G, name + ".safetensors"), "pt") as f:
        meta = f.metadata()
        for k in f.keys():
            blob[k] = f.get_tensor(k)
    return blob, eval(meta["config"])


def test_vjepa_oracle_matches_reference_model():
    blob, c = load("vjepa_tiny")
    cfg = VJEPAConfig(**c)
    p_on = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    p_tg = {k[7:]: v for k, v in blob.items() if k.startswith("tparam.")}
    assert set(p_on) == set(param_shapes(cfg))
    pix, cm, tm = blob["in.pixel_values_videos"], blob["in.context_mask"], blob["in.target_mask"]
    out = model_forward(p_on, cfg, pix, [cm], [tm])
    assert torch.allclose(out["last_hidden_state"], blob["out.last_hidden_state"], rtol=1e-4, atol=2e-5)
    assert torch.allclose(out["predictor_output"], blob["out.predictor"], rtol=1e-4, atol=2e-5)
    loss, pred, tgt, grads = vjepa_step(p_on, p_tg, cfg, pix, [cm], [tm])
    assert torch.allclose(tgt, blob["out.target_hidden_state"], rtol=1e-4, atol=2e-5)
    assert abs(float(loss) - float(blob["out.loss"])) <= 1e-5 * float(blob["out.loss"])
    for k, g in grads.items():
        ref = blob["grad." + k]
        assert float((g - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-9, k
    # shapes asserted by the reference's tests/test_vjepa_model.py:69-76,122-125
    assert out["masked_hidden_state"].shape == (2, cm.shape[1], cfg.hidden_size) and out["predictor_output"].shape == (2, tm.shape[1], cfg.hidden_size)
    assert model_forward(p_on, cfg, pix, [cm], [tm], skip_predictor=True)["predictor_output"] is None


def test_rotate_uses_tiled_angles_not_interleaved():
    # Appendix A.8 / SURVEY V1: cos/sin are tiled ([f0..f_{D/2-1}] twice) while the rotated pairs are interleaved
    x = torch.arange(8.0).view(1, 1, 1, 8)
    pos = torch.tensor([3.0])
    y = rotate(x, pos)
    om = 1.0 / 10000 ** (torch.arange(4.0) / 4.0)
    ang = torch.cat([3.0 * om, 3.0 * om])
    rot = torch.tensor([-1.0, 0.0, -3.0, 2.0, -5.0, 4.0, -7.0, 6.0])
    assert torch.allclose(y.view(-1), x.view(-1) * ang.cos() + rot * ang.sin())


def test_rope3d_segments():
    q = torch.randn(1, 2, 5, 32)
    out = rope3d(q, torch.tensor([0, 1, 4, 17, 63]), grid_size=4)
    assert torch.equal(out[..., 30:], q[..., 30:])          # 32 -> 10/10/10 + 2 passthrough
    assert torch.allclose(out[:, :, 0], q[:, :, 0])          # position 0: identity


def mask_cases():
    z = np.load(os.path.join(G, "vjepa_masks.npz"))
    for k in z.files:
        if k.endswith("_cfg"):
            ci = int(k.split("_")[1])
            yield z, ci, z[k]


@pytest.mark.parametrize("case", list(mask_cases()), ids=lambda c: f"case{c[1]}")
def test_vjepa_mask_oracle_bit_exact(case):
    z, ci, cfgv = case
    inp, ps, nb = tuple(int(v) for v in cfgv[:3]), tuple(int(v) for v in cfgv[3:6]), int(cfgv[6])
    if inp[0] * inp[1] * inp[2] // (ps[0] * ps[1] * ps[2]) > 5000:
        pytest.skip("large grid: python loops")
    gen = VJEPAMaskGeneratorOracle(inp, ps, (0.2, 0.8), (0.3, 3.0), nb)
    for seed in range(4):
        torch.manual_seed(seed)
        for draw in range(2):
            ctx, tgt = gen()
            key = f"vmask_{ci}_seed{seed}_{draw}"
            assert np.array_equal(ctx.reshape(-1).numpy(), z[key + "_ctx"]) and np.array_equal(tgt.reshape(-1).numpy(), z[key + "_tgt"])
            # reference tests/test_vjepa_transforms.py:106-107: partition, disjoint
            n = (inp[0] // ps[0]) * (inp[1] // ps[1]) * (inp[2] // ps[2])
            assert ctx.numel() + tgt.numel() == n and not set(ctx.reshape(-1).tolist()) & set(tgt.reshape(-1).tolist())


def test_ema_update():
    cfg = VJEPAConfig(patch_size=8, crop_size=16, frames_per_clip=4, tubelet_size=2, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                      pred_hidden_size=64, pred_num_hidden_layers=1, pred_num_attention_heads=2)
    a, b = init_params(cfg, 0), init_params(cfg, 1)
    out = ema_update(a, b, 0.9)
    k = "encoder.layer.0.mlp.fc1.weight"
    assert torch.allclose(out[k], 0.9 * a[k] + 0.1 * b[k])


def test_vjepa_oracle_matches_reference_model_with_mask_lists():
    """Two (context, target) mask pairs (tests/golden/vjepa_multimask.safetensors, written by make_golden.py from the reference VJEPA2Model): the case
    the reference's tests/test_vjepa_model.py:96-125 exercises - same shapes, plus values and gradients."""
    blob, c = load("vjepa_multimask")
    cfg = VJEPAConfig(**c)
    p = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    leaf = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    cms, tms = list(blob["in.context_mask"]), list(blob["in.target_mask"])
    out = model_forward(leaf, cfg, blob["in.pixel_values_videos"], cms, tms)
    assert out["last_hidden_state"].shape == (2, 32, 64) and out["masked_hidden_state"].shape == (4, 20, 64)
    assert out["predictor_output"].shape == (4, 12, 64) and out["target_hidden_state"].shape == out["predictor_output"].shape
    for k, name in (("last_hidden_state", "out.last_hidden_state"), ("masked_hidden_state", "out.masked_hidden_state"),
                    ("target_hidden_state", "out.target_hidden_state"), ("predictor_output", "out.predictor")):
        assert torch.allclose(out[k], blob[name], rtol=1e-4, atol=2e-5), k
    (out["predictor_output"] * blob["in.w"]).sum().backward()
    for k, v in leaf.items():
        ref = blob["grad." + k]
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((g - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-8, k
