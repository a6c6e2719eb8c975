"""The attention kernels behind transformers' attention registry (smb_vision_amd/attention_plugin.py): the reference's models pick their
attention by name from ALL_ATTENTION_FUNCTIONS (src/models/vjepa/modeling_vjepa.py:345-364, src/models/videomae/modeling_videomae.py:270-289).

Checked against (a) transformers' own eager attention function on the same operands in fp32 and (b) the installed
transformers.VideoMAEForPreTraining - the class src/run_mim.py:13-21 imports - run once with "eager" and once with "smbv" attention, same
weights, same inputs, bf16 autocast as the Trainer sets it.  Tolerances: bf16 operands and bf16 P / dS inside the kernels
(tests/test_ops_gpu.py: 1.5e-2 max / 6e-3 rms forward, 2.5e-2 / 1e-2 backward).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach(), b.detach()
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


def rms_err(a, b):
    a, b = a.detach(), b.detach()
    return float((a.float() - b.float()).pow(2).mean().sqrt() / (b.float().pow(2).mean().sqrt() + 1e-30))


@pytest.mark.parametrize("B,H,S,Dh,dtype", [(2, 3, 333, 64, torch.bfloat16), (1, 4, 200, 32, torch.bfloat16), (1, 2, 130, 64, torch.float32)])
def test_registered_function_matches_eager_attention(B, H, S, Dh, dtype):
    from transformers.models.videomae.modeling_videomae import eager_attention_forward
    from smb_vision_amd.attention_plugin import smbv_attention
    g = torch.Generator(device="cpu").manual_seed(5)
    q, k, v = ((torch.randn(B, H, S, Dh, generator=g).cuda().to(dtype)).requires_grad_(True) for _ in range(3))
    module = torch.nn.Module().train()
    out, probs = smbv_attention(module, q, k, v, None, scaling=Dh ** -0.5, dropout=0.0, is_causal=False)
    assert probs is None and out.shape == (B, S, H, Dh) and out.dtype == dtype and out.is_contiguous()
    qf, kf, vf = (t.detach().to(torch.bfloat16).float().requires_grad_(True) for t in (q, k, v))
    ref, _ = eager_attention_forward(module, qf, kf, vf, None, scaling=Dh ** -0.5, dropout=0.0)
    assert rel_err(out, ref) < 1.5e-2 and rms_err(out, ref) < 6e-3
    d_out = torch.randn(B, S, H, Dh, generator=g).cuda().to(torch.bfloat16)
    out.backward(d_out.to(dtype))
    ref.backward(d_out.float())
    for name, a, b in (("q", q, qf), ("k", k, kf), ("v", v, vf)):
        assert a.grad.shape == a.shape and a.grad.dtype == dtype
        assert rel_err(a.grad, b.grad) < 2.5e-2 and rms_err(a.grad, b.grad) < 1e-2, name


def test_hf_videomae_with_smbv_attention_matches_eager():
    """The reference's model class, unmodified, with only the attention swapped by name."""
    from transformers import VideoMAEConfig, VideoMAEForPreTraining
    from smb_vision_amd import attention_plugin
    name = attention_plugin.register()
    kw = dict(image_size=64, patch_size=16, num_channels=1, num_frames=64, tubelet_size=16, hidden_size=192, num_hidden_layers=2, num_attention_heads=3,
              intermediate_size=384, decoder_num_attention_heads=3, decoder_hidden_size=96, decoder_num_hidden_layers=1, decoder_intermediate_size=192,
              norm_pix_loss=True)
    torch.manual_seed(0)
    eager = VideoMAEForPreTraining(VideoMAEConfig(**kw, attn_implementation="eager")).cuda().train()
    ours = VideoMAEForPreTraining(VideoMAEConfig(**kw, attn_implementation=name)).cuda().train()
    ours.load_state_dict(eager.state_dict())
    assert ours.videomae.encoder.layer[0].attention.attention.config._attn_implementation == name
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.rand(2, 64, 1, 64, 64, generator=g).cuda()
    n_tokens = (64 // 16) * (64 // 16) * (64 // 16)
    mask = torch.zeros(2, n_tokens, dtype=torch.bool)
    for b in range(2):
        mask[b, torch.randperm(n_tokens, generator=g)[:40]] = True
    mask = mask.cuda()
    outs = []
    for m in (eager, ours):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o = m(pixel_values=x, bool_masked_pos=mask)
        o.loss.backward()
        outs.append(o)
    l0, l1 = float(outs[0].loss.detach()), float(outs[1].loss.detach())
    assert abs(l0 - l1) < 5e-3 * abs(l0)
    assert rms_err(outs[1].logits, outs[0].logits) < 2e-2
    pe, po = dict(eager.named_parameters()), dict(ours.named_parameters())
    # (a key bias shifts every score of a row by the same amount: its gradient is zero up to rounding noise - skipped)
    worst = max((rms_err(po[n].grad, pe[n].grad), n) for n in pe if pe[n].grad is not None and not n.endswith("key.bias"))
    # encoder layer 0 (S = 24 visible tokens, Dh 64) and the decoder layer (S = 64, Dh 32) both run the HIP kernels; bf16 autocast noise of the two runs
    # differs only inside attention
    assert worst[0] < 3e-2, worst


def test_unsupported_arguments_raise():
    from smb_vision_amd.attention_plugin import smbv_attention
    q = torch.zeros(1, 2, 16, 64, device="cuda", dtype=torch.bfloat16)
    m = torch.nn.Module()
    with pytest.raises(ValueError):
        smbv_attention(m, q, q, q, torch.zeros(1, 1, 16, 16, device="cuda"))
    with pytest.raises(ValueError):
        smbv_attention(m, q, q, q, None, dropout=0.1)
    with pytest.raises(ValueError):
        smbv_attention(m, q, q, q, None, is_causal=True)
    with pytest.raises(ValueError):
        smbv_attention(m, q[..., :48], q[..., :48], q[..., :48], None)
