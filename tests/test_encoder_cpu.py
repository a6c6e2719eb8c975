"""Encoder-only forwards: the CPU restatement (oracle/mim_oracle.py encoder_forward / classification_forward)
against goldens produced by the reference's model classes (tests/golden/make_golden.py make_encoder), and the host-side surface of the
drop-in classes (state-dict names, errors).  No GPU."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import mim_oracle as mo

G = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
           intermediate_size=256)


@pytest.fixture(scope="module")
def blob():
    return load_file(os.path.join(G, "encoder_tiny.safetensors"))


def _params(blob, pre):
    return {k[len(pre):]: v for k, v in blob.items() if k.startswith(pre)}


def ocfg():
    return mo.MIMConfig(decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1, decoder_intermediate_size=64, **CFG)


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_oracle_matches_reference_encoder_outputs(blob):
    p = _params(blob, "cls.param.")
    pix, mask = blob["in.pixel_values"], blob["in.bool_masked_pos"].bool()
    with torch.no_grad():
        assert rel(mo.encoder_forward(p, ocfg(), pix), blob["cls.out.last_hidden_state"]) < 1e-5
        assert rel(mo.encoder_forward(p, ocfg(), pix, mask), blob["cls.out.visible_hidden_state"]) < 1e-5
        assert rel(mo.classification_forward(p, ocfg(), pix), blob["cls.out.logits"]) < 1e-5
        pe = _params(blob, "enc.param.")
        assert rel(mo.encoder_forward(pe, ocfg(), pix, None, prefix="", final_ln=True), blob["enc.out.last_hidden_state"]) < 1e-5


def test_oracle_hidden_states_match_the_reference_classes(blob):
    """output_hidden_states=True (spec :452-457, 478-483): the oracle's per-layer residual stream against tests/golden/hidden_states_tiny.safetensors
    (make_golden.py make_hidden_states: the installed transformers classes on this fixture's weights; they reproduced its stored outputs bit for bit)."""
    hs = load_file(os.path.join(G, "hidden_states_tiny.safetensors"))
    p, pe = _params(blob, "cls.param."), _params(blob, "enc.param.")
    pix, mask = blob["in.pixel_values"], blob["in.bool_masked_pos"].bool()
    with torch.no_grad():
        for tag, params, m, prefix, fl in (("cls.all", p, None, "videomae.", False), ("cls.visible", p, mask, "videomae.", False), ("enc.all", pe, None, "", True)):
            got = []
            mo.encoder_forward(params, ocfg(), pix, m, prefix=prefix, final_ln=fl, hidden_states=got)
            assert len(got) == CFG["num_hidden_layers"] + 1
            for l, h in enumerate(got):
                assert h.shape == hs[f"{tag}.{l}"].shape and rel(h, hs[f"{tag}.{l}"]) < 1e-5, (tag, l)
        # the pre-training model hands on its ENCODER's hidden states (visible tokens; spec :900-907)
        mim = load_file(os.path.join(G, "mim_tiny_p16.safetensors"))
        mc = mo.MIMConfig(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=256,
                          decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1, decoder_intermediate_size=128)
        got = []
        mo.encoder_forward(_params(mim, "param."), mc, mim["in.pixel_values"], mim["in.bool_masked_pos"].bool(), hidden_states=got)
        for l, h in enumerate(got):
            assert rel(h, hs[f"mim.{l}"]) < 1e-5, l


def test_oracle_fine_tune_gradients_match_reference(blob):
    p = {k: v.clone().requires_grad_(True) for k, v in _params(blob, "cls.param.").items()}
    logits = mo.classification_forward(p, ocfg(), blob["in.pixel_values"])
    loss = torch.nn.functional.cross_entropy(logits, blob["in.labels"])
    assert float(loss) == pytest.approx(float(blob["cls.out.loss"]), rel=1e-5)
    loss.backward()
    for k, g in _params(blob, "cls.grad.").items():
        assert rel(p[k].grad, g) < 2e-4, k


def test_drop_in_class_surface(blob):
    from transformers import VideoMAEConfig
    from smb_vision_amd import _lib
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEForVideoClassification, SmbVideoMAEModel
    m = SmbVideoMAEForVideoClassification(VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **CFG))
    assert set(dict(m.named_parameters())) == set(_params(blob, "cls.param."))
    e0 = SmbVideoMAEModel(VideoMAEConfig(num_channels=1, use_mean_pooling=False, **CFG))
    assert set(dict(e0.get_input_embeddings().named_parameters())) == {"projection.weight", "projection.bias"}   # spec :525-526
    assert not m.load_state_dict(_params(blob, "cls.param."), strict=True).missing_keys
    e = SmbVideoMAEModel(VideoMAEConfig(num_channels=1, use_mean_pooling=False, **CFG))
    assert set(dict(e.named_parameters())) == set(_params(blob, "enc.param."))
    hf5 = {}
    for k, v in _params(blob, "enc.param.").items():   # a transformers-5 checkpoint: query/key/value.bias
        if k.endswith("q_bias"):
            hf5[k.replace("q_bias", "query.bias")] = v
            hf5[k.replace("q_bias", "key.bias")] = torch.zeros_like(v)
        elif k.endswith("v_bias"):
            hf5[k.replace("v_bias", "value.bias")] = v
        else:
            hf5[k] = v
    assert not e.load_state_dict(hf5, strict=True).missing_keys
    with pytest.raises(_lib.SmbvError, match="no CPU fallback"):
        with torch.no_grad():
            e(torch.zeros(1, 32, 1, 64, 64))
    with pytest.raises(ValueError, match="single-channel"):
        SmbVideoMAEModel(VideoMAEConfig(num_channels=3, **CFG))
