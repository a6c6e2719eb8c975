"""CPU: pin the oracle (oracle/) against the golden vectors produced by the reference's own classes in the build
container (tests/golden/make_golden.py): the reference MaskGenerator and transformers.VideoMAEForPreTraining."""
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle import mask_oracle
from oracle.mim_oracle import MIMConfig, init_params, labels_from_pixels, mim_forward, mim_step, param_shapes, sinusoid_table

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    blob = {}
    with safe_open(os.path.join(G, name + ".safetensors"), "pt") as f:
        meta = f.metadata()
        for k in f.keys():
            blob[k] = f.get_tensor(k)
    return blob, eval(meta["config"]), eval(meta["mask"])


def mask_cases():
    z = np.load(os.path.join(G, "masks.npz"))
    for k in z.files:
        if k.endswith("_a"):
            _, sz, dp, mps, ps, ratio, seed = k[:-2].split("_")
            yield z, k[:-2], int(sz), int(dp), int(mps), int(ps), float(ratio), int(seed[4:])


@pytest.mark.parametrize("case", list(mask_cases()), ids=lambda c: c[1])
def test_mask_oracle_bit_exact(case):
    z, key, sz, dp, mps, ps, ratio, seed = case
    n, nm = z[key + "_n"]
    if n > 4096 and seed > 1:
        pytest.skip("large grid: seeds 0,1 only (pure-python loops)")
    np.random.seed(seed)
    for suffix in ("_a", "_b"):
        m = mask_oracle.generate_mask(sz, dp, mps, ps, ratio)
        ref = np.unpackbits(z[key + suffix])[:n].astype(bool)
        assert m.shape == (n,) and int(m.sum()) == nm
        assert np.array_equal(m, ref)


def test_mask_counts_match_survey():
    # 20,480 tokens, 13,312 masked / 7,168 visible at 512x512x320, ratio 0.65, either mask patch size
    for mps in (16, 32):
        rs, rd, scale, tc, mc = mask_oracle.mask_counts(512, 320, mps, 16, 0.65)
        assert tc * scale ** 3 == 20480 and mc * scale ** 3 == 13312
    with pytest.raises(ValueError):
        mask_oracle.mask_counts(100, 32, 16, 16, 0.5)


def test_sinusoid_table_matches_loop_definition():
    # the reference builds the table with a python double loop (modeling_videomae.py:95-106)
    n, d = 7, 10
    ref = np.array([[p / np.power(10000, 2 * (j // 2) / d) for j in range(d)] for p in range(n)])
    ref[:, 0::2] = np.sin(ref[:, 0::2])
    ref[:, 1::2] = np.cos(ref[:, 1::2])
    assert torch.equal(sinusoid_table(n, d), torch.FloatTensor(ref))


@pytest.mark.parametrize("name", ["mim_tiny_p16", "mim_small_p8"])
def test_oracle_matches_reference_model(name):
    blob, c, _ = load_golden(name)
    cfg = MIMConfig(**c)
    params = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    assert set(params) == set(param_shapes(cfg))
    pix, mask = blob["in.pixel_values"], blob["in.bool_masked_pos"].bool()
    loss, logits, grads = mim_step(params, cfg, pix, mask)
    assert abs(float(loss) - float(blob["out.loss"])) <= 1e-5 * abs(float(blob["out.loss"]))
    assert torch.allclose(logits, blob["out.logits"], rtol=1e-4, atol=2e-5)
    for k, g in grads.items():
        ref = blob["grad." + k]
        scale = float(ref.abs().max()) + 1e-12
        assert float((g - ref).abs().max()) <= 1e-4 * scale + 1e-9, k


def test_init_params_is_the_golden_weight_set():
    blob, c, _ = load_golden("mim_tiny_p16")
    p = init_params(MIMConfig(**c), seed=42, nonzero_small=True)
    for k, v in p.items():
        assert torch.equal(v, blob["param." + k]), k


def test_labels_constant_brick_is_zero():
    cfg = MIMConfig(image_size=16, num_frames=16, patch_size=8, tubelet_size=8, hidden_size=64, num_attention_heads=1)
    pix = torch.full((1, 16, 1, 16, 16), 0.25)
    assert float(labels_from_pixels(pix, cfg).abs().max()) == 0.0


def test_bf16_emulation_stays_close():
    blob, c, _ = load_golden("mim_small_p8")
    cfg = MIMConfig(**c)
    params = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    loss, logits = mim_forward(params, cfg, blob["in.pixel_values"], blob["in.bool_masked_pos"].bool(), emulate_bf16=True)
    assert abs(float(loss) - float(blob["out.loss"])) < 5e-3 * float(blob["out.loss"])
