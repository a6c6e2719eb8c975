"""Encoder-only forwards on the HIP path (smbv_encoder_forward through the drop-in classes) against the reference goldens: embedding
export over all tokens, over the visible tokens, with the final LayerNorm, the classification logits, and model.videomae(image) on the
pre-training model (src/run_inference.py:78-83).  Tolerances: bf16 operands / fp32 accumulation as everywhere."""
import os

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
           intermediate_size=256)


@pytest.fixture(scope="module")
def blob():
    return load_file(os.path.join(G, "encoder_tiny.safetensors"))


def _params(blob, pre):
    return {k[len(pre):]: v for k, v in blob.items() if k.startswith(pre)}


def close(got, ref, tol=2e-2):
    got, ref = got.float().cpu(), ref.float()
    assert float((got - ref).abs().max() / ref.abs().max()) < tol
    assert float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()) < tol / 2


def test_classification_and_embedding_export(blob):
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEForVideoClassification
    m = SmbVideoMAEForVideoClassification(VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **CFG))
    m.load_state_dict(_params(blob, "cls.param."), strict=True)
    m = m.cuda().eval()
    pix, mask = blob["in.pixel_values"].cuda(), blob["in.bool_masked_pos"].bool().cuda()
    with torch.no_grad():
        out = m(pixel_values=pix)
        # logits are O(0.1) sums of +-O(1) LayerNorm'ed features times 0.05-scale weights: the bf16 noise of the features is not averaged down
        close(out.logits, blob["cls.out.logits"], tol=5e-2)
        close(m.videomae(pix).last_hidden_state, blob["cls.out.last_hidden_state"])
        close(m.videomae(pix, bool_masked_pos=mask).last_hidden_state, blob["cls.out.visible_hidden_state"])
        labels = torch.tensor([0, 2]).cuda()
        ref_loss = torch.nn.functional.cross_entropy(blob["cls.out.logits"], labels.cpu())
        assert float(m(pixel_values=pix, labels=labels).loss) == pytest.approx(float(ref_loss), rel=3e-2)


def test_fine_tune_step_gradients(blob):
    """Classification loss -> backward through smbv_encoder_backward: every parameter gradient against the reference's."""
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEForVideoClassification
    m = SmbVideoMAEForVideoClassification(VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **CFG))
    m.load_state_dict(_params(blob, "cls.param."), strict=True)
    m = m.cuda().train()
    out = m(pixel_values=blob["in.pixel_values"].cuda(), labels=blob["in.labels"].cuda())
    assert float(out.loss) == pytest.approx(float(blob["cls.out.loss"]), rel=1e-2)
    out.loss.backward()
    ref = _params(blob, "cls.grad.")
    worst = 0.0
    for k, p in m.named_parameters():
        g, r = p.grad.float().cpu(), ref[k].float()
        err = float((g - r).norm() / (r.norm() + 1e-12))
        worst = max(worst, err)
        assert err < 8e-2, (k, err)   # per-tensor ||dg|| <= 8e-2 ||g||: bf16 operands, tiny-magnitude logits gradient
    assert worst > 0  # (gradients are not trivially zero)
    # a second step reuses the arena and the workspace
    m.zero_grad(set_to_none=True)
    out2 = m(pixel_values=blob["in.pixel_values"].cuda(), labels=blob["in.labels"].cuda())
    out2.loss.backward()
    k0 = "videomae.encoder.layer.0.attention.attention.query.weight"
    assert torch.allclose(dict(m.named_parameters())[k0].grad.cpu(), ref[k0], atol=float(ref[k0].abs().max()) * 0.1)


def test_model_with_final_layernorm(blob):
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEModel
    e = SmbVideoMAEModel(VideoMAEConfig(num_channels=1, use_mean_pooling=False, **CFG))
    e.load_state_dict(_params(blob, "enc.param."), strict=True)
    e = e.cuda().eval()
    with torch.no_grad():
        out = e(blob["in.pixel_values"].cuda())
    assert out.last_hidden_state.shape == (2, 32, 128)
    close(out.last_hidden_state, blob["enc.out.last_hidden_state"])


def test_output_hidden_states_of_every_class_match_the_reference(blob):
    """output_hidden_states=True (spec :452-457, 478-483, 596-599, 900-907): the embedding output and every layer's output, NOT the final LayerNorm's - out of the
    step workspace (smbv_encoder_hidden_states / smbv_mim_hidden_states, ABI 3006) against tests/golden/hidden_states_tiny.safetensors (the installed transformers
    classes on this fixture's weights).  Inference, fine-tuning (autograd on), by config, as a tuple, through model.videomae(...), the pre-training model in its
    three modes; the outputs beside them are unchanged."""
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEForVideoClassification, SmbVideoMAEModel
    hs = load_file(os.path.join(G, "hidden_states_tiny.safetensors"))
    pix, mask = blob["in.pixel_values"].cuda(), blob["in.bool_masked_pos"].bool().cuda()

    def check(got, tag, n=3):
        assert isinstance(got, tuple) and len(got) == n
        for l, h in enumerate(got):
            assert h.dtype == torch.float32 and not h.requires_grad and tuple(h.shape) == tuple(hs[f"{tag}.{l}"].shape)
            close(h, hs[f"{tag}.{l}"])

    m = SmbVideoMAEForVideoClassification(VideoMAEConfig(num_channels=1, num_labels=3, use_mean_pooling=True, **CFG))
    m.load_state_dict(_params(blob, "cls.param."), strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        plain = m(pixel_values=pix)
        assert plain.hidden_states is None
        out = m(pixel_values=pix, output_hidden_states=True)
        check(out.hidden_states, "cls.all")
        assert torch.equal(out.logits, plain.logits)          # the per-layer workspace form computes what the shared-set form computes
        check(m.videomae(pix, output_hidden_states=True).hidden_states, "cls.all")
        vis = m.videomae(pix, bool_masked_pos=mask, output_hidden_states=True)
        check(vis.hidden_states, "cls.visible")
        close(vis.last_hidden_state, blob["cls.out.visible_hidden_state"])
        tup = m(pixel_values=pix, output_hidden_states=True, return_dict=False)
        assert len(tup) == 2 and torch.equal(tup[0], plain.logits)
        check(tup[1], "cls.all")
        assert m.videomae(pix).hidden_states is None          # and back to the inference workspace
    # fine-tuning: autograd on, the hidden states are detached copies, the backward still runs
    m.train()
    out = m(pixel_values=pix, labels=blob["in.labels"].cuda(), output_hidden_states=True)
    check(out.hidden_states, "cls.all")
    out.loss.backward()
    g = dict(m.named_parameters())["videomae.encoder.layer.0.attention.attention.query.weight"].grad
    r = blob["cls.grad.videomae.encoder.layer.0.attention.attention.query.weight"]
    assert float((g.cpu() - r).norm() / r.norm()) < 8e-2
    # resolved from the config when the argument is None (spec :596-599); the final LayerNorm is not part of the tuple
    cfg = VideoMAEConfig(num_channels=1, use_mean_pooling=False, output_hidden_states=True, **CFG)
    e = SmbVideoMAEModel(cfg)
    e.load_state_dict(_params(blob, "enc.param."), strict=True)
    e = e.cuda().eval()
    with torch.no_grad():
        out = e(pix)
        check(out.hidden_states, "enc.all")
        close(out.last_hidden_state, blob["enc.out.last_hidden_state"])
        assert e(pix, output_hidden_states=False).hidden_states is None
        with pytest.raises(NotImplementedError):
            e(pix, output_attentions=True)
    # the pre-training model hands on its encoder's hidden states over the visible tokens
    mim = load_file(os.path.join(G, "mim_tiny_p16.safetensors"))
    mc = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=256,
              decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1, decoder_intermediate_size=128)
    mpix, mmask = mim["in.pixel_values"].cuda(), mim["in.bool_masked_pos"].bool().cuda()
    for extra in ({}, {"smbv_recompute": "always"}):
        pm = SmbVideoMAEForPreTraining(VideoMAEConfig(num_channels=1, **mc, **extra))
        pm.load_state_dict(_params(mim, "param."), strict=True)
        pm = pm.cuda().train()
        if extra:
            pm.gradient_checkpointing_enable()
        out = pm(pixel_values=mpix, bool_masked_pos=mmask, output_hidden_states=True)
        check(out.hidden_states, "mim")
        assert float(out.loss) == pytest.approx(float(mim["out.loss"]), rel=5e-3)
        out.loss.backward()                                    # the backward reads the same workspace the copies were taken from
        k0 = "videomae.encoder.layer.0.attention.attention.query.weight"
        assert float((dict(pm.named_parameters())[k0].grad.cpu() - mim["grad." + k0]).norm() / mim["grad." + k0].norm()) < 2.5e-2
        with torch.no_grad():
            tup = pm(pixel_values=mpix, bool_masked_pos=mmask, output_hidden_states=True, return_dict=False)
        assert len(tup) == 3
        check(tup[2], "mim")
        assert pm(pixel_values=mpix, bool_masked_pos=mmask).hidden_states is None
        with pytest.raises(NotImplementedError):
            pm(pixel_values=mpix, bool_masked_pos=mmask, output_attentions=True)


def test_pretraining_model_videomae_call_matches_its_own_encoder(blob):
    """model.videomae(image) on SmbVideoMAEForPreTraining: same encoder weights as the classification golden -> same embeddings."""
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    cfg = VideoMAEConfig(num_channels=1, decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1,
                         decoder_intermediate_size=128, **CFG)
    m = SmbVideoMAEForPreTraining(cfg)
    enc = {k: v for k, v in _params(blob, "cls.param.").items() if k.startswith("videomae.")}
    missing = m.load_state_dict(enc, strict=False)
    assert all(not k.startswith("videomae.") for k in missing.missing_keys)
    m = m.cuda().eval()
    with torch.no_grad():
        close(m.videomae(blob["in.pixel_values"].cuda()).last_hidden_state, blob["cls.out.last_hidden_state"])


@pytest.mark.parametrize("task,extra", [("classification", 0), ("regression", 2), ("multilabel_classification", 0)])
def test_hf_trainer_fine_tunes(tmp_path, task, extra):
    from smb_vision_amd.run_classification import main
    metrics = main(["--output_dir", str(tmp_path), "--image_size", "32", "--depth", "16", "--patch_size", "8", "--task_type", task, "--num_labels", "3",
                    "--num_additional_features", str(extra), "--hidden_size", "128", "--num_hidden_layers", "2", "--num_attention_heads", "2",
                    "--intermediate_size", "256", "--max_steps", "6", "--per_device_train_batch_size", "2", "--report_to", "none", "--bf16", "true",
                    "--learning_rate", "1e-3", "--logging_steps", "1", "--save_strategy", "no", "--dataloader_num_workers", "0"])
    assert 0 < metrics["train_loss"] < 3.0
    assert os.path.exists(os.path.join(tmp_path, "model.safetensors"))
