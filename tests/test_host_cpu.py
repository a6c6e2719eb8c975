"""CPU: host logic of the product package — mask generator, index split, model-class surface (constructor, state-dict
names, save/load in both layouts, errors), C-ABI library loads and exports every symbol of include/smbv.h."""
import ctypes
import os
import re
import tempfile

import numpy as np
import pytest
import torch

from oracle.mim_oracle import MIMConfig, param_shapes
from smb_vision_amd import _lib
from smb_vision_amd.masking import MaskGenerator, split_mask_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
TINY = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=64, num_hidden_layers=2, num_attention_heads=1,
            intermediate_size=256, decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1,
            decoder_intermediate_size=128)


def test_library_exports_every_declared_symbol():
    _lib.load()  # raises if the .so is missing or lacks a bound symbol
    hdr = open(os.path.join(ROOT, "include", "smbv.h")).read()
    declared = set(re.findall(r"\b(smbv_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libsmbv.so does not export {name}"
    assert declared == set(_lib.EXPORTS)
    assert lib.smbv_version() >= 1000


def test_bad_config_is_rejected_without_gpu():
    lib = _lib.load()
    cfg = _lib.MimCfg()
    assert lib.smbv_mim_workspace_bytes(ctypes.byref(cfg)) == 0
    assert "mim" in _lib.last_error()


def test_unsupported_config_options_raise_instead_of_being_ignored():
    """Options the HIP path does not compute are refused at construction (a drop-in that silently trained without dropout would not be one); head widths:
    multiples of 8 up to 64 are accepted by the C side (64 / 32 native, others zero-padded), anything else is SMBV_ERR_SHAPE."""
    from transformers import VideoMAEConfig, VJEPA2Config
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    from smb_vision_amd.modeling_videomae_encoder import SmbVideoMAEModel
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    tiny = dict(num_channels=1, image_size=32, num_frames=16, patch_size=16, tubelet_size=16, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                intermediate_size=64, decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1, decoder_intermediate_size=64)
    for bad in (dict(hidden_dropout_prob=0.1), dict(attention_probs_dropout_prob=0.1), dict(hidden_act="relu"), dict(qkv_bias=False), dict(num_channels=3)):
        with pytest.raises(ValueError):
            SmbVideoMAEForPreTraining(VideoMAEConfig(**{**tiny, **bad}))
        with pytest.raises(ValueError):
            SmbVideoMAEModel(VideoMAEConfig(**{**tiny, **bad}))
    vj = dict(patch_size=8, crop_size=16, frames_per_clip=4, tubelet_size=2, in_chans=1, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
              pred_hidden_size=64, pred_num_hidden_layers=1, pred_num_attention_heads=1)
    for bad in (dict(attention_dropout=0.1), dict(attention_probs_dropout_prob=0.1), dict(drop_path_rate=0.1), dict(in_chans=3)):
        with pytest.raises(ValueError):
            SmbVJEPA2Model(VJEPA2Config(**{**vj, **bad}))
    lib = _lib.load()
    ok = dict(B=1, T=32, H=32, W=32, ps=16, ts=16, N=8, Nvis=4, Nmask=4, ff=64, layers=1, dff=64, dlayers=1, ln_eps=1e-12, dec_norm_eps=1e-5)
    for d, heads, fine in ((64, 1, True), (64, 2, True), (96, 2, True), (64, 4, True), (160, 2, False), (72, 6, False), (64, 3, False)):
        cfg = _lib.MimCfg(d=d, heads=heads, dd=64, dheads=1, **ok)
        n = lib.smbv_mim_workspace_bytes(ctypes.byref(cfg))
        assert (n > 0) == fine, (d, heads, n, _lib.last_error())


def test_workspace_size_vit_base():
    lib = _lib.load()
    cfg = _lib.MimCfg(B=1, T=320, H=512, W=512, ps=16, ts=16, N=20480, Nvis=7168, Nmask=13312, d=768, heads=12, ff=3072, layers=12,
                      dd=384, dheads=6, dff=1536, dlayers=4, ln_eps=1e-12, dec_norm_eps=1e-5)
    n = lib.smbv_mim_workspace_bytes(ctypes.byref(cfg))
    assert 3e9 < n < 12e9


def mask_cases():
    z = np.load(os.path.join(G, "masks.npz"))
    for k in z.files:
        if k.endswith("_a"):
            _, sz, dp, mps, ps, ratio, seed = k[:-2].split("_")
            yield z, k[:-2], int(sz), int(dp), int(mps), int(ps), float(ratio), int(seed[4:])


@pytest.mark.parametrize("case", list(mask_cases()), ids=lambda c: c[1])
def test_product_mask_generator_bit_exact(case):
    z, key, sz, dp, mps, ps, ratio, seed = case
    n, nm = z[key + "_n"]
    gen = MaskGenerator(input_size=sz, depth=dp, mask_patch_size=mps, model_patch_size=ps, mask_ratio=ratio)
    assert gen.num_tokens == n and gen.num_masked == nm
    np.random.seed(seed)
    for suffix in ("_a", "_b"):
        m = gen()
        assert m.dtype == torch.bool and m.shape == (n,)
        assert np.array_equal(m.numpy(), np.unpackbits(z[key + suffix])[:n].astype(bool))


def test_mask_generator_errors():
    with pytest.raises(ValueError, match="Input size"):
        MaskGenerator(input_size=100, depth=32, mask_patch_size=16)
    with pytest.raises(ValueError, match="Depth"):
        MaskGenerator(input_size=64, depth=40, mask_patch_size=16)
    with pytest.raises(ValueError, match="Mask patch size"):
        MaskGenerator(input_size=96, depth=96, mask_patch_size=24, model_patch_size=16)


def test_split_mask_indices_matches_boolean_indexing():
    g = torch.Generator().manual_seed(0)
    B, N, nm = 3, 64, 40
    mask = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(N, generator=g)[:nm]] = True
    vis, msk = split_mask_indices(mask, nm)
    ar = torch.arange(N).expand(B, -1)
    assert torch.equal(vis.long(), ar[~mask].reshape(B, -1)) and torch.equal(msk.long(), ar[mask].reshape(B, -1))
    assert vis.dtype == torch.int32


@pytest.fixture(scope="module")
def model_and_cfg():
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    cfg = VideoMAEConfig(num_channels=1, **TINY)
    torch.manual_seed(0)
    return SmbVideoMAEForPreTraining(cfg), cfg


def test_state_dict_names_are_the_reference_layout(model_and_cfg):
    model, _ = model_and_cfg
    sd = model.state_dict()
    ref = param_shapes(MIMConfig(**TINY))
    assert set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == ref[k] for k in ref)
    assert model.base_model_prefix == "videomae" and model.main_input_name == "pixel_values" and model.supports_gradient_checkpointing
    # _init_weights semantics
    assert float(sd["mask_token"].abs().sum()) == 0 and float(sd["videomae.encoder.layer.0.attention.attention.q_bias"].abs().sum()) == 0
    assert float(sd["decoder.norm.weight"].min()) == 1.0
    assert 0.015 < float(sd["decoder.head.weight"].std()) < 0.025


def test_param_count_vit_base():
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import param_table
    cfg = VideoMAEConfig(image_size=512, patch_size=16, num_channels=1, num_frames=320, tubelet_size=16)
    n = sum(int(np.prod(s)) for _, s, _ in param_table(cfg))
    # SURVEY.md §8(a15) counts 97,171,840 with the transformers-5 layout, whose attention carries a (softmax-invariant) key bias;
    # the pinned-era layout this class uses has none: 12*768 + 4*384 fewer.
    assert n == 97_171_840 - (12 * 768 + 4 * 384)


def test_save_load_roundtrip_and_hf5_layout(model_and_cfg):
    from transformers import VideoMAEForPreTraining
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    model, cfg = model_and_cfg
    with tempfile.TemporaryDirectory() as d:
        model.save_pretrained(d)
        m2 = SmbVideoMAEForPreTraining.from_pretrained(d, config=cfg)
        for (k, a), b in zip(model.state_dict().items(), m2.state_dict().values()):
            assert torch.equal(a, b), k
    hf = VideoMAEForPreTraining(cfg)
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if n.endswith("bias"):
                p.normal_()
    with tempfile.TemporaryDirectory() as d:
        hf.save_pretrained(d)
        m3 = SmbVideoMAEForPreTraining.from_pretrained(d, config=cfg)
    hsd, sd = hf.state_dict(), m3.state_dict()
    a = "videomae.encoder.layer.1.attention.attention."
    assert torch.equal(sd[a + "q_bias"], hsd[a + "query.bias"]) and torch.equal(sd[a + "v_bias"], hsd[a + "value.bias"])
    assert torch.equal(sd["decoder.head.weight"], hsd["decoder.head.weight"])
    # plain load_state_dict accepts it too
    model.load_state_dict(hsd)


def test_errors_mirror_the_reference(model_and_cfg):
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    model, cfg = model_and_cfg
    mask = torch.zeros(1, 32, dtype=torch.bool)
    with pytest.raises(ValueError, match="channel dimension"):
        model(torch.zeros(1, 32, 3, 64, 64), mask)
    with pytest.raises(ValueError, match="doesn't match model"):
        model(torch.zeros(1, 32, 1, 32, 32), mask)
    with pytest.raises(ValueError, match="boolean mask"):
        model(torch.zeros(1, 32, 1, 64, 64), None)
    with pytest.raises(_lib.SmbvError, match="no CPU fallback"):
        model(torch.zeros(1, 32, 1, 64, 64), mask)  # the product path never falls back to the CPU
    with pytest.raises(ValueError):
        SmbVideoMAEForPreTraining(VideoMAEConfig(num_channels=1, norm_pix_loss=False, **TINY))
    model.gradient_checkpointing_enable()
    assert model.gradient_checkpointing


def vmask_cases():
    z = np.load(os.path.join(G, "vjepa_masks.npz"))
    for k in z.files:
        if k.endswith("_cfg"):
            yield z, int(k.split("_")[1]), z[k]


@pytest.mark.parametrize("case", list(vmask_cases()), ids=lambda c: f"case{c[1]}")
def test_product_vjepa_mask_generator_bit_exact(case):
    from smb_vision_amd.masking import VJEPAMaskGenerator
    z, ci, cfgv = case
    inp, ps, nb = tuple(int(v) for v in cfgv[:3]), tuple(int(v) for v in cfgv[3:6]), int(cfgv[6])
    gen = VJEPAMaskGenerator(inp, ps, (0.2, 0.8), (0.3, 3.0), nb)
    for seed in range(4):
        torch.manual_seed(seed)
        for draw in range(2):
            d = gen({})
            key = f"vmask_{ci}_seed{seed}_{draw}"
            assert np.array_equal(d["context_mask"].reshape(-1).numpy(), z[key + "_ctx"])
            assert np.array_equal(d["target_mask"].reshape(-1).numpy(), z[key + "_tgt"])


def test_vjepa_model_class_surface():
    from transformers import VJEPA2Config
    from oracle.vjepa_oracle import VJEPAConfig, param_shapes as vshapes
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    c = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
             mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)
    m = SmbVJEPA2Model(VJEPA2Config(**c))
    ref = vshapes(VJEPAConfig(**c))
    # names, shapes AND parameters() order (MomentumEncoder.update zips the two models' parameters, run_vjepa.py:97)
    assert [n for n, _ in m.named_parameters()] == list(ref)
    assert all(tuple(p.shape) == ref[n] for n, p in m.named_parameters())
    assert m.main_input_name == "pixel_values_videos" and m.base_model_prefix == "vjepa2"
    sd = m.state_dict()
    assert float(sd["predictor.embeddings.mask_tokens"].abs().sum()) == 0 and float(sd["encoder.layernorm.weight"].min()) == 1.0
    with tempfile.TemporaryDirectory() as d:
        m.save_pretrained(d)
        m2 = SmbVJEPA2Model.from_pretrained(d)
        assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with pytest.raises(_lib.SmbvError, match="no CPU fallback"):
        m(pixel_values_videos=torch.zeros(1, 8, 1, 32, 32))


def test_host_index_lists_equal_the_device_sort_and_the_reference_boolean_indexing():
    from smb_vision_amd.masking import MaskGenerator, split_mask_indices, split_mask_indices_host
    np.random.seed(3)
    gen = MaskGenerator(128, 64, 32, 16, 0.6)
    mask = torch.stack([gen() for _ in range(3)])
    vis, msk = split_mask_indices_host(mask)
    v2, m2 = split_mask_indices(mask, int(mask[0].sum()))
    assert vis.dtype == torch.int32 and torch.equal(vis, v2) and torch.equal(msk, m2)
    ar = torch.arange(mask.shape[1]).expand_as(mask)
    assert torch.equal(ar[~mask].reshape(3, -1), vis.long()) and torch.equal(ar[mask].reshape(3, -1), msk.long())   # spec :134-137 / :811-812 order
    bad = mask.clone()
    bad[1, :] = False
    with pytest.raises(ValueError, match="same number"):
        split_mask_indices_host(bad)


def test_decay_groups_equal_hf_split_of_the_reference_classes():
    """SmbTrainer's decay / no-decay split must be the one HF computes for the reference classes (LayerNorm gains and every bias undecayed);
    tests/golden/decay_names.json holds that split (make_golden.py decay)."""
    import json
    from transformers import VideoMAEConfig, VJEPA2Config
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    from smb_vision_amd.trainer import SmbTrainer
    ref = json.load(open(os.path.join(G, "decay_names.json")))
    vj = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
              mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)
    for key, model in (("mim", SmbVideoMAEForPreTraining(VideoMAEConfig(num_channels=1, **TINY))), ("vjepa", SmbVJEPA2Model(VJEPA2Config(**vj)))):
        names = [n for n, _ in model.named_parameters()]
        decay = set(SmbTrainer.get_decay_parameter_names(None, model))
        assert sorted(n for n in names if n in decay) == ref[key]["decay"], key
        assert sorted(n for n in names if n not in decay) == ref[key]["no_decay"], key


def test_cfg_struct_size_guard_and_abi_major():
    """a binding compiled against another struct layout is refused (struct_bytes), and the loader checks the ABI major."""
    lib = _lib.load()
    assert lib.smbv_version() // 1000 == _lib.ABI_MAJOR
    hdr = open(os.path.join(ROOT, "include", "smbv.h")).read()
    assert int(re.search(r"#define SMBV_ABI_MAJOR (\d+)", hdr).group(1)) == _lib.ABI_MAJOR
    cfg = _lib.MimCfg(B=1, T=320, H=512, W=512, ps=16, ts=16, N=20480, Nvis=7168, Nmask=13312, d=768, heads=12, ff=3072, layers=12,
                      dd=384, dheads=6, dff=1536, dlayers=4, ln_eps=1e-12, dec_norm_eps=1e-5)
    assert cfg.struct_bytes == ctypes.sizeof(_lib.MimCfg) and lib.smbv_mim_workspace_bytes(ctypes.byref(cfg)) > 0
    cfg.struct_bytes -= 8  # the round-2 layout without recompute / attn_fp8
    assert lib.smbv_mim_workspace_bytes(ctypes.byref(cfg)) == 0 and "struct_bytes" in _lib.last_error()
    v = _lib.VjepaCfg()
    v.struct_bytes = 4
    assert lib.smbv_vjepa_workspace_bytes(ctypes.byref(v)) == 0 and "struct_bytes" in _lib.last_error()
    e = _lib.EncoderCfg()
    e.struct_bytes = 0
    assert lib.smbv_encoder_workspace_bytes(ctypes.byref(e)) == 0 and "struct_bytes" in _lib.last_error()


def test_attention_plugin_registers_and_refuses_cpu():
    """smb_vision_amd/attention_plugin.py: the name lands in transformers' attention registry (the reference's dispatch point,
    src/models/vjepa/modeling_vjepa.py:345-364) and a model built with it fails loudly without a GPU - no CPU fallback."""
    import torch
    from transformers import VideoMAEConfig, VideoMAEForPreTraining
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    from smb_vision_amd import attention_plugin
    name = attention_plugin.register()
    assert ALL_ATTENTION_FUNCTIONS[name] is attention_plugin.smbv_attention
    cfg = VideoMAEConfig(image_size=32, patch_size=16, num_channels=1, num_frames=32, tubelet_size=16, hidden_size=128, num_hidden_layers=1,
                         num_attention_heads=2, intermediate_size=256, decoder_num_attention_heads=2, decoder_hidden_size=128,
                         decoder_num_hidden_layers=1, decoder_intermediate_size=256, attn_implementation=name)
    model = VideoMAEForPreTraining(cfg)
    mask = torch.zeros(1, 8, dtype=torch.bool)
    mask[0, :5] = True
    import pytest
    with pytest.raises(RuntimeError, match="MI355X"):
        model(pixel_values=torch.rand(1, 32, 1, 32, 32), bool_masked_pos=mask)


def test_vjepa_collate_ships_the_index_arrays():
    """run_vjepa.collate_fn(num_patches=N): the step's index arithmetic runs on the host (modeling_vjepa.index_plumbing) and travels with the batch;
    the arrays are what the device-side derivation gives (same function, other device) and a permutation where they must be one."""
    import torch
    from smb_vision_amd.modeling_vjepa import index_plumbing
    from smb_vision_amd.run_vjepa import collate_fn
    g = torch.Generator().manual_seed(3)
    N = 96
    perm = torch.randperm(N, generator=g)
    cm, tm = perm[:30].sort().values, perm[30:80].sort().values
    examples = [{"image": torch.rand(2, 1, 4, 4), "context_mask": cm, "target_mask": tm} for _ in range(3)]
    batch = collate_fn(examples, num_patches=N)
    assert set(batch) == {"pixel_values_videos", "context_mask", "target_mask", "smbv_plumbing_context", "smbv_plumbing_target"}
    assert set(collate_fn(examples)) == {"pixel_values_videos", "context_mask", "target_mask"}   # the reference's batch when not asked for more
    pc, pt = batch["smbv_plumbing_context"], batch["smbv_plumbing_target"]
    assert pc["enc_needed"] == 30 and pt["enc_needed"] == 50 and pc["n_masks"] == 1
    for pl, sel in ((pc, cm), (pt, tm)):
        assert pl["all_idx"].dtype == torch.int32 and pl["all_idx"].shape == (3, N)
        assert torch.equal(pl["all_idx"].sort(dim=1).values, torch.arange(N, dtype=torch.int32).repeat(3, 1))      # a permutation of the tokens
        assert torch.equal(pl["all_idx"][:, :sel.numel()].long(), sel.repeat(3, 1))                                  # consumed tokens first, ascending
    # predictor slots: targets first (by position), then the context
    assert torch.equal(pc["pred_pos"][0, :50].long(), tm) and torch.equal(pc["pred_pos"][0, 50:].long(), cm)
    assert torch.equal(pc["tgt_slot"][0].long(), torch.arange(50)) and torch.equal(pc["ctx_slot"][0].long(), 50 + torch.arange(30))
    ref = index_plumbing([batch["context_mask"][0].long()], [batch["target_mask"][0].long()], N, "context")
    assert all(torch.equal(pc[k], ref[k]) for k in ("tgt_idx", "pred_pos", "ctx_slot", "tgt_slot", "all_idx", "need_rows", "ctx_idx"))


def test_synthetic_datasets_give_one_fixed_item_per_index_and_leave_the_global_rng_alone():
    """Round 4 (resume): HF skips batches by index when it resumes, without producing the items - the harness datasets therefore derive an item's mask from its
    index (what the reference's PersistentDataset caches with the image), not from however many items were produced before, and leave the global streams the
    mask generators draw from exactly as they found them."""
    from smb_vision_amd.masking import MaskGenerator, VJEPAMaskGenerator
    from smb_vision_amd.run_mim import SyntheticVolumeDataset
    from smb_vision_amd.run_vjepa import SyntheticVJEPADataset
    ds = SyntheticVolumeDataset(8, 32, 16, MaskGenerator(32, 16, 16, 16, 0.5), seed=7)
    np.random.seed(123)
    expect_next = np.random.RandomState(123).permutation(5)
    a3 = ds[3]
    assert np.array_equal(np.random.permutation(5), expect_next)          # the global numpy stream did not move
    _ = ds[0], ds[5]
    b3 = ds[3]
    assert torch.equal(a3["mask"], b3["mask"]) and torch.equal(a3["image"], b3["image"])
    assert not torch.equal(ds[3]["mask"], ds[4]["mask"]) or not torch.equal(ds[3]["image"], ds[4]["image"])
    gen = VJEPAMaskGenerator(input_size=(16, 32, 32), patch_size=(8, 8, 8), pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=2)
    dv = SyntheticVJEPADataset(4, 32, 16, gen, 11)
    torch.manual_seed(5)
    state = torch.random.get_rng_state()
    x2 = dv[2]
    assert torch.equal(torch.random.get_rng_state(), state)               # nor the global torch stream
    _ = dv[0]
    y2 = dv[2]
    assert torch.equal(x2["context_mask"], y2["context_mask"]) and torch.equal(x2["target_mask"], y2["target_mask"])


def test_resume_helpers_follow_the_reference_scripts(tmp_path):
    """src/run_mim.py:263-276: the newest checkpoint-* of output_dir is what a run resumes from unless --resume_from_checkpoint names another; the metric
    helper is the reference's compute_metrics (loss = first prediction) made safe for more than one evaluation batch."""
    import types
    from transformers import EvalPrediction
    from smb_vision_amd.trainer import detect_last_checkpoint, keep_loss_only, loss_only_metrics
    args = types.SimpleNamespace(output_dir=str(tmp_path / "run"), resume_from_checkpoint=None, do_train=True)
    assert detect_last_checkpoint(args) is None                           # no directory yet
    for n in (2, 10, 4):
        (tmp_path / "run" / f"checkpoint-{n}").mkdir(parents=True)
    assert detect_last_checkpoint(args).endswith("checkpoint-10")
    args.do_train = False                                                  # src/run_mim.py:265: only a training run looks for checkpoints
    assert detect_last_checkpoint(args) is None
    args.do_train = True
    args.overwrite_output_dir = True                                       # (where the installed transformers still has the flag)
    assert detect_last_checkpoint(args) is None
    assert loss_only_metrics(EvalPrediction(predictions=(np.array([1.0, 3.0]), np.zeros((2, 4))), label_ids=None)) == {"loss": 2.0}
    assert loss_only_metrics(EvalPrediction(predictions=np.array([0.5]), label_ids=None)) == {"loss": 0.5}
    kept = keep_loss_only((torch.tensor(0.25), torch.zeros(2, 3, 4)), None)
    assert kept.shape == (1,) and float(kept[0]) == 0.25


def test_collate_functions_unwrap_singleton_lists_as_the_reference_does(tmp_path):
    """src/run_mim.py:195-202 / src/run_vjepa.py:145-152 ("common in MONAI/PyTorch datasets"): items arrive as ``item``, ``[item]`` or ``[[item]]``; a longer
    list is NOT peeled (the reference's loop stops at len != 1).  Also: items that carry a cached-volume record instead of an image travel on as raw volumes."""
    from smb_vision_amd import volcache as vc
    from smb_vision_amd.feed import RAW_KEY
    from smb_vision_amd.run_mim import collate_fn as mim_collate
    from smb_vision_amd.run_vjepa import collate_fn as vj_collate
    gen = MaskGenerator(32, 16, 16, 16, 0.5)
    np.random.seed(0)
    items = [{"image": torch.rand(16, 1, 32, 32, generator=torch.Generator().manual_seed(i)), "mask": gen()} for i in range(3)]
    plain = mim_collate(items)
    wrapped = mim_collate([[items[0]], [[items[1]]], (items[2],)])
    assert set(plain) == {"pixel_values", "bool_masked_pos", "visible_index", "masked_index"}
    assert all(torch.equal(plain[k], wrapped[k]) for k in plain)
    assert plain["pixel_values"].shape == (3, 16, 1, 32, 32) and plain["bool_masked_pos"].dtype == torch.bool
    with pytest.raises((TypeError, KeyError)):
        mim_collate([[items[0], items[1]]])                                # a two-element list is not an item
    one = mim_collate([[items[1]]])
    assert torch.equal(one["pixel_values"][0], items[1]["image"]) and one["pixel_values"].shape[0] == 1

    cm, tm = torch.arange(0, 5), torch.arange(5, 8)
    vitems = [{"image": torch.rand(2, 1, 4, 4), "context_mask": cm, "target_mask": tm} for _ in range(2)]
    a, b = vj_collate(vitems), vj_collate([[vitems[0]], [[vitems[1]]]])
    assert torch.equal(a["pixel_values_videos"], b["pixel_values_videos"]) and torch.equal(a["context_mask"][0], b["context_mask"][0])

    # raw cached volumes: the record travels, not the voxels
    for i in range(2):
        vc.write_volume(str(tmp_path / f"v{i}.smbvol"), np.full((8, 8, 4), i, dtype=np.int16))
    ds = vc.CachedVolumeDataset(str(tmp_path), mask_generator=gen, mask_seed=3)
    assert len(ds) == 2 and torch.equal(ds[1]["mask"], ds[1]["mask"]) and isinstance(ds[0]["volume"], vc.VolumeRecord)
    raw = mim_collate([[ds[0]], ds[1]])
    assert set(raw) == {RAW_KEY, "bool_masked_pos", "visible_index", "masked_index"} and [r.path for r in raw[RAW_KEY]] == ds.paths
    with pytest.raises(FileNotFoundError):
        vc.CachedVolumeDataset(str(tmp_path / "missing"))


def test_eval_only_invocation_neither_trains_nor_looks_for_checkpoints(tmp_path):
    """ADVICE r4 / src/run_mim.py:265, 439-455: `--do_eval true` without `--do_train` evaluates only."""
    import types
    from smb_vision_amd.trainer import train_and_evaluate

    class T:
        eval_dataset = [0]
        calls = []

        def train(self, **kw):
            self.calls.append("train")
            return types.SimpleNamespace(metrics={"train_loss": 1.0})

        def evaluate(self):
            self.calls.append("evaluate")
            return {"eval_loss": 0.5}

        def __getattr__(self, name):   # save_model, log_metrics, save_metrics, save_state
            return lambda *a, **k: self.calls.append(name)

    t = T()
    m = train_and_evaluate(t, types.SimpleNamespace(do_train=False, do_eval=True, resume_from_checkpoint=None))
    assert m == {"eval_loss": 0.5} and "train" not in t.calls and "save_model" not in t.calls and "evaluate" in t.calls
    t.calls.clear()
    m = train_and_evaluate(t, types.SimpleNamespace(do_train=True, do_eval=True, resume_from_checkpoint=None))
    assert t.calls[0] == "train" and "save_model" in t.calls and m == {"train_loss": 1.0, "eval_loss": 0.5}
