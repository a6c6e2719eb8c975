"""GPU parity of the V-JEPA path (SmbVJEPA2Model -> smbv_vjepa_forward/backward, smbv_rope3d, smbv_l1_*, smbv_ema_update) against the
goldens of the reference classes (tests/golden/vjepa_tiny.safetensors) and the CPU oracle.  Tolerances as for MIM (bf16 operands,
fp32 accumulation / LayerNorm / softmax / loss): outputs max 3e-2 of scale, loss 1e-2 rel, per-tensor gradient ||d|| <= 8e-2 ||g||
(L1's sign() gradient makes individual elements flip on ties, so the gradient bound is a little looser than for MSE)."""
import ctypes as C
import os

import pytest
import torch
from safetensors import safe_open

from oracle.vjepa_oracle import VJEPAConfig, rope3d

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    blob = {}
    with safe_open(os.path.join(G, name + ".safetensors"), "pt") as f:
        meta = f.metadata()
        for k in f.keys():
            blob[k] = f.get_tensor(k)
    return blob, eval(meta["config"])


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30))


@pytest.mark.parametrize("Dh,hs", [(64, 64), (32, 64), (16, 64)])
def test_rope3d_matches_oracle(Dh, hs):
    from smb_vision_amd import _lib
    from smb_vision_amd.modeling_vjepa import rope_table
    lib = _lib.load()
    B, S, H, grid = 2, 48, 3, 4
    wa = H * hs
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * S, 3 * wa, generator=g).bfloat16()
    ids = torch.stack([torch.randperm(64, generator=g)[:S].sort().values for _ in range(B)])
    table = rope_table(Dh, 8, "cuda")
    x = qkv.cuda().clone()
    pos = ids.to(torch.int32).cuda().contiguous()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.smbv_rope3d(x.data_ptr(), 3 * wa, wa, B * S, S, H, hs, Dh, grid, pos.data_ptr(), table.data_ptr(), 0, st), "rope")
    for sec in (0, 1):
        src = qkv.float()[:, sec * wa:(sec + 1) * wa].view(B, S, H, hs)[..., :Dh].permute(0, 2, 1, 3)
        ref = rope3d(src, ids, grid).permute(0, 2, 1, 3)
        got = x.float().cpu()[:, sec * wa:(sec + 1) * wa].view(B, S, H, hs)[..., :Dh]
        assert rel(got, ref) < 1e-2
    assert torch.equal(x.cpu()[:, 2 * wa:], qkv[:, 2 * wa:])  # v untouched
    seg = 2 * ((Dh // 3) // 2)
    assert torch.equal(x.cpu().view(B * S, 3, H, hs)[:, :2, :, 3 * seg:], qkv.view(B * S, 3, H, hs)[:, :2, :, 3 * seg:])  # pass-through tail of every head slot
    # transpose map == autograd of the forward map
    src = qkv.float()[:, :wa].view(B, S, H, hs)[..., :Dh].permute(0, 2, 1, 3).clone().requires_grad_(True)
    gy = torch.randn(B, H, S, Dh, generator=g)
    rope3d(src, ids, grid).backward(gy)
    gbuf = torch.zeros(B * S, 3 * wa)
    gbuf[:, :wa].view(B, S, H, hs)[..., :Dh].copy_(gy.permute(0, 2, 1, 3))
    gd = gbuf.bfloat16().cuda()
    _lib.check(lib.smbv_rope3d(gd.data_ptr(), 3 * wa, wa, B * S, S, H, hs, Dh, grid, pos.data_ptr(), table.data_ptr(), 1, st), "rope^T")
    got = gd.float().cpu()[:, :wa].view(B, S, H, hs)[..., :Dh].permute(0, 2, 1, 3)
    assert rel(got, src.grad) < 1.5e-2


@pytest.mark.parametrize("M,w,H,Dh,hs,S,with_pos", [
    (2048, 768, 12, 64, 64, 1024, True),     # the encoder's product: N = 2304, K = 768 - the ring kernel's rotary epilogue
    (1000, 384, 12, 32, 32, 500, False),     # the predictor's: head_dim 32 (a 4-column group straddles two axes: seg = 10), positions = row % S, ragged M
    (1536, 384, 6, 64, 64, 512, True),       # N = 1152, K = 384
    (300, 256, 3, 40, 64, 100, True),        # padded head slots; N = 576: not a ring shape - the product runs plain with the rope3d launch behind it
])
def test_qkv_product_with_rotary_epilogue_is_the_two_launches_bitwise(M, w, H, Dh, hs, S, with_pos):
    """smbv_gemm_nt_rope (ABI 3004): the fused QKV product whose epilogue applies the 3-axis rotary map to the q | k sections of the bf16-rounded product
    (modeling_vjepa.py:204-228, :311-326) leaves bitwise the matrix that smbv_gemm_nt followed by smbv_rope3d leaves - which test_rope3d_matches_oracle
    pins against the oracle."""
    from smb_vision_amd import ops
    from smb_vision_amd.modeling_vjepa import rope_table
    grid = 8
    wa = H * hs
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, w, generator=g).bfloat16().cuda()
    Wt = (torch.randn(3 * wa, w, generator=g) / w ** 0.5).bfloat16().cuda()
    bias = (0.1 * torch.randn(3 * wa, generator=g)).cuda()
    pos = torch.randint(0, grid * grid * 6, (M,), generator=g).to(torch.int32).cuda() if with_pos else None
    rope = dict(table=rope_table(Dh, 16, "cuda"), pos=pos, S=S, grid=grid, Dh=Dh, head_stride=hs, wa=wa)   # frame / height / width ids < 16
    fused = ops.gemm_nt(A, Wt, bias=bias, rope=rope)["bf16"]
    two = ops.gemm_nt(A, Wt, bias=bias)["bf16"]
    plain = two.clone()
    ops.rope3d_(two, wa, H, rope)
    torch.cuda.synchronize()
    assert torch.equal(fused, two)
    unknown = ops.gemm_nt(A, Wt, bias=bias, rope=dict(rope, table_rows=0))["bf16"]   # a table of unknown extent is not staged into LDS: the two launches
    assert torch.equal(unknown, two)
    assert not torch.equal(fused[:, :2 * wa], plain[:, :2 * wa]) and torch.equal(fused[:, 2 * wa:], plain[:, 2 * wa:])   # q | k rotated, v untouched
    from smb_vision_amd._lib import SmbvError
    with pytest.raises(SmbvError):   # an epilogue the rotary form does not combine with
        ops.gemm_nt(A, Wt, bias=bias, rope=rope, out_f32=True)


def test_l1_and_ema_kernels():
    from smb_vision_amd.modeling_vjepa import MomentumEncoder, l1_loss
    torch.manual_seed(7)
    p = torch.randn(3, 50, 64, device="cuda", requires_grad=True)
    t = torch.randn(3, 50, 64, device="cuda")
    loss = l1_loss(p, t)
    (loss * 0.5).backward()
    pr = p.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.l1_loss(pr, t)
    (ref * 0.5).backward()
    assert abs(float(loss) - float(ref)) < 1e-6 * float(ref) + 1e-7
    assert torch.allclose(p.grad, pr.grad, rtol=1e-6, atol=1e-9)
    q = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(s, device="cuda")) for s in [(70000,), (5, 3), (64,)]])
    k = torch.nn.ParameterList([torch.nn.Parameter(torch.randn_like(x)) for x in q])
    want = [0.9 * b.detach() + 0.1 * a.detach() for a, b in zip(q, k)]
    MomentumEncoder(k, 0.9).update(q)
    for w, b in zip(want, k):
        # operands reach |x| ~ 4.5 and the sum cancels: one rounding of each product is up to ulp(4)/2 = 2.4e-7 absolute whatever the result's size
        assert torch.allclose(b, w, rtol=1e-6, atol=1e-6)


def build(c, params):
    from transformers import VJEPA2Config
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    m = SmbVJEPA2Model(VJEPA2Config(**c))
    m.load_state_dict(params, strict=True)
    return m.cuda().train()


def test_vjepa_step_matches_reference_golden():
    from smb_vision_amd.modeling_vjepa import MomentumEncoder, make_target_encoder, vjepa_loss
    blob, c = load("vjepa_tiny")
    p_on = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    p_tg = {k[7:]: v for k, v in blob.items() if k.startswith("tparam.")}
    model = build(c, p_on)
    target = make_target_encoder(build(c, p_tg))
    pix = blob["in.pixel_values_videos"].cuda()
    cm, tm = blob["in.context_mask"].cuda(), blob["in.target_mask"].cuda()
    inputs = {"pixel_values_videos": pix, "context_mask": [cm], "target_mask": [tm]}
    loss, out = vjepa_loss(model, target, None, inputs)   # (the harness path: last encoder blocks restricted to the consumed rows, smbv_needed)
    loss.backward()
    torch.cuda.synchronize()
    assert out.last_hidden_state is None and out.target_hidden_state is None
    with torch.no_grad():
        full = model(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm])   # the reference's forward: every output
    assert rel(full.last_hidden_state.cpu(), blob["out.last_hidden_state"]) < 3e-2
    with torch.no_grad():   # get_vision_features (modeling_vjepa.py:1165-1167): no masks, the encoder over every token - the same last_hidden_state
        feats = model.get_vision_features(pix)
    assert feats.shape == full.last_hidden_state.shape and rel(feats.cpu(), blob["out.last_hidden_state"]) < 3e-2
    assert dict(model.get_input_embeddings().named_parameters()).keys() == {"proj_3d.weight", "proj_3d.bias"}
    assert rel(full.predictor_output.last_hidden_state.cpu(), blob["out.predictor"]) < 3e-2
    assert rel(out.masked_hidden_state.cpu(), full.masked_hidden_state.cpu()) < 2e-3
    assert rel(out.predictor_output.last_hidden_state.detach().cpu(), blob["out.predictor"]) < 3e-2
    with torch.no_grad():
        tgt = target(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm], skip_predictor=True)
    assert tgt.predictor_output is None and rel(tgt.target_hidden_state.cpu(), blob["out.target_hidden_state"]) < 3e-2
    assert out.masked_hidden_state.shape == (2, cm.shape[1], 128) and out.predictor_output.last_hidden_state.shape == (2, tm.shape[1], 128)
    assert abs(float(loss) - float(blob["out.loss"])) < 1e-2 * float(blob["out.loss"])
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        ref = blob["grad." + k]
        n = float(ref.norm())
        g = p.grad.cpu()
        if n < 1e-10:
            assert float(g.norm()) < 1e-6, k
            continue
        r = float((g - ref).norm()) / n
        worst = max(worst, (k, r), key=lambda t: t[1])
        # key biases: without RoPE a key bias is softmax-invariant (zero gradient); with RoPE only the position-dependent
        # rotation leaks a small, cancellation-dominated gradient, so bf16 noise weighs more against its norm
        tol = 2e-1 if k.endswith("attention.key.bias") else 8e-2
        assert r < tol, (k, r, n)
    print("vjepa loss", float(loss), "ref", float(blob["out.loss"]), "worst grad", worst)
    # EMA after the loss (run_vjepa.py:139): every target parameter moves towards the online one
    before = [p.detach().clone() for p in target.parameters()]
    MomentumEncoder(target, 0.5).update(model)
    for b, t, q in zip(before, target.parameters(), model.parameters()):
        assert torch.allclose(t, 0.5 * b + 0.5 * q.detach(), rtol=1e-6, atol=1e-7)


def test_vjepa_padded_head_widths_match_oracle():
    """Head widths with no native attention build (encoder 64 / 4 = 16, predictor 96 / 2 = 48) run zero-padded in 64-wide slots (block.h):
    one step against the CPU oracle on the same weights (parity unpinned by a reference golden: the goldens' widths are 64 and 32)."""
    import oracle.vjepa_oracle as vo
    from smb_vision_amd.modeling_vjepa import make_target_encoder, vjepa_loss
    c = dict(TINY, hidden_size=64, num_attention_heads=4, pred_hidden_size=96, pred_num_attention_heads=2)
    cfg = vo.VJEPAConfig(**c)
    params = vo.init_params(cfg, seed=5, nonzero_small=True)
    n = (c["frames_per_clip"] // c["tubelet_size"]) * (c["crop_size"] // c["patch_size"]) ** 2
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
    cm, tm = perm[: n // 3].sort().values[None].repeat(2, 1), perm[n // 3:].sort().values[None].repeat(2, 1)
    pix = torch.rand(2, c["frames_per_clip"], 1, c["crop_size"], c["crop_size"], generator=torch.Generator().manual_seed(4))
    model = build(c, params)
    target = make_target_encoder(model)
    loss, out = vjepa_loss(model, target, None, {"pixel_values_videos": pix.cuda(), "context_mask": [cm.cuda()], "target_mask": [tm.cuda()]})
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_pred, _, ref_grads = vo.vjepa_step(params, params, cfg, pix, [cm], [tm])
    assert rel(out.predictor_output.last_hidden_state.detach().cpu(), ref_pred) < 3e-2
    assert abs(float(loss) - float(ref_loss)) < 1e-2 * float(ref_loss)
    for k, p_ in model.named_parameters():
        ref = ref_grads[k]
        nrm = float(ref.norm())
        if nrm < 1e-10:
            assert float(p_.grad.norm()) < 1e-6, k
            continue
        r = float((p_.grad.cpu() - ref).norm()) / nrm
        assert r < (2e-1 if k.endswith("attention.key.bias") else 8e-2), (k, r, nrm)


def test_vjepa_default_masks_and_errors():
    blob, c = load("vjepa_tiny")
    p_on = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    model = build(c, p_on).eval()
    pix = blob["in.pixel_values_videos"].cuda()
    with torch.no_grad():
        out = model(pixel_values_videos=pix)       # reference tests/test_vjepa_model.py:69-76: default forward -> (B, N, hidden)
    assert out.last_hidden_state.shape == (2, 64, 128) and out.predictor_output.last_hidden_state.shape == (2, 64, 128)
    with pytest.raises(ValueError):
        model(pixel_values_videos=torch.zeros(1, 8, 1, 16, 16).cuda())
    with pytest.raises(Exception):
        model(pixel_values_videos=pix.cpu())


def test_hf_trainer_runs_vjepa(tmp_path):
    from smb_vision_amd.run_vjepa import main
    metrics = main(["--output_dir", str(tmp_path), "--image_size", "32", "--depth", "16", "--patch_size", "8", "--num_mask_blocks", "2",
                    "--hidden_size", "128", "--num_hidden_layers", "2", "--num_attention_heads", "2", "--pred_hidden_size", "64",
                    "--pred_num_hidden_layers", "2", "--pred_num_attention_heads", "2", "--max_steps", "6", "--per_device_train_batch_size", "2",
                    "--report_to", "none", "--bf16", "true", "--learning_rate", "1e-3", "--logging_steps", "1", "--save_strategy", "no",
                    "--max_grad_norm", "1.0", "--dataloader_num_workers", "0"])
    assert 0 < metrics["train_loss"] < 2.0
    assert os.path.exists(os.path.join(tmp_path, "model.safetensors"))


def test_hf_trainer_vjepa_resumes_and_evaluates(tmp_path):
    """run_vjepa.py:191-205, 258-275: auto-resume + evaluate.  Not bit-identical to an uninterrupted run by the reference's own design: the EMA target
    encoder lives on the trainer, is not checkpointed, and a resumed run re-copies it from the restored online weights."""
    import json
    import shutil
    from smb_vision_amd.run_vjepa import main
    common = ["--image_size", "32", "--depth", "16", "--patch_size", "8", "--num_mask_blocks", "2", "--hidden_size", "128", "--num_hidden_layers", "2",
              "--num_attention_heads", "2", "--pred_hidden_size", "64", "--pred_num_hidden_layers", "2", "--pred_num_attention_heads", "2",
              "--per_device_train_batch_size", "2", "--report_to", "none", "--bf16", "true", "--learning_rate", "1e-3", "--logging_steps", "1",
              "--max_grad_norm", "1.0", "--dataloader_num_workers", "0", "--save_strategy", "steps", "--save_steps", "2"]
    a, b = tmp_path / "a", tmp_path / "b"
    main(["--output_dir", str(a), "--max_steps", "4", *common])
    os.makedirs(b)
    shutil.copytree(a / "checkpoint-2", b / "checkpoint-2")
    m = main(["--output_dir", str(b), "--max_steps", "4", "--do_train", "true", "--do_eval", "true", "--per_device_eval_batch_size", "2", *common])
    state = json.load(open(b / "trainer_state.json"))
    assert state["global_step"] == 4
    assert [h["step"] for h in state["log_history"] if "loss" in h][:4] == [1, 2, 3, 4]    # steps 1-2 restored from the checkpoint's history, 3-4 run here
    assert 0 < m["eval_loss"] < 2.0


TINY = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
            mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)


def _vjepa_dp_worker(rank, world, port, q, overlap=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from transformers import VJEPA2Config
        from smb_vision_amd.dp import DataParallelEngine
        from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
        torch.cuda.set_device(0)
        torch.manual_seed(0)
        m = SmbVJEPA2Model(VJEPA2Config(**TINY)).cuda().train()
        # gloo has no device-side AVG: the engine's SUM + scale path, as in tests/test_dp_cpu.py; overlap=True adds the per-group HIP events the
        # backward records and the communication stream waits for (what the RCCL launch uses)
        dp = DataParallelEngine(m, overlap=overlap, bucket_bytes=1 << 16)
        dp.broadcast_parameters(m)
        g = torch.Generator().manual_seed(10 + rank)      # different data per rank
        pix = torch.rand(1, TINY["frames_per_clip"], 1, TINY["crop_size"], TINY["crop_size"], generator=g).cuda()
        n = (TINY["frames_per_clip"] // TINY["tubelet_size"]) * (TINY["crop_size"] // TINY["patch_size"]) ** 2
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(5))
        cm, tm = perm[: n // 2].sort().values[None].cuda(), perm[n // 2:].sort().values[None].cuda()
        out = m(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm])
        out.predictor_output.last_hidden_state.square().mean().backward()
        q.put((rank, m.flat_grads().double().sum().item(), m.flat_grads().abs().double().sum().item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_vjepa_gradients_are_averaged_across_ranks(overlap):
    """Two processes on the one GPU (gloo): after backward both ranks hold the same, non-trivial, averaged gradient arena."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_vjepa_dp_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == pytest.approx(res[1][1], rel=1e-12, abs=1e-12) and res[0][2] == pytest.approx(res[1][2], rel=1e-12) and res[0][2] > 0


def test_vjepa_mask_lists_match_the_reference_shapes_and_the_oracle():
    """Lists of masks (reference tests/test_vjepa_model.py:96-125: two context and two target masks -> (4, 20, 64) / (4, 12, 64) at batch 2,
    32 tokens): apply_masks concatenates the pairs along the batch, the predictor runs on every (volume, pair) sequence, and the encoder
    receives the SUM of the pairs' gradients.  Shapes as the reference test asserts them; values and every gradient against the oracle."""
    from oracle.vjepa_oracle import VJEPAConfig, model_forward
    blob, c = load("vjepa_multimask")          # inputs, weights and the reference VJEPA2Model's own outputs / gradients for this case
    cfg = VJEPAConfig(**c)
    assert cfg.num_patches == 32
    params = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    pix, w = blob["in.pixel_values_videos"], blob["in.w"]
    cms, tms = list(blob["in.context_mask"]), list(blob["in.target_mask"])
    model = build(c, params)
    out = model(pixel_values_videos=pix.cuda(), context_mask=[m.cuda() for m in cms], target_mask=[m.cuda() for m in tms])
    assert out.last_hidden_state.shape == (2, 32, 64)
    assert out.masked_hidden_state.shape == (4, 20, 64)
    assert out.predictor_output.last_hidden_state.shape == (4, 12, 64)
    assert out.target_hidden_state.shape == out.predictor_output.last_hidden_state.shape
    (out.predictor_output.last_hidden_state * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ref = model_forward(leaf, cfg, pix, cms, tms)
    (ref["predictor_output"] * w).sum().backward()
    assert rel(out.last_hidden_state.cpu(), ref["last_hidden_state"].detach()) < 3e-2
    assert rel(out.predictor_output.last_hidden_state.detach().cpu(), ref["predictor_output"].detach()) < 3e-2
    assert rel(out.masked_hidden_state.cpu(), ref["masked_hidden_state"].detach()) < 3e-2
    assert rel(out.target_hidden_state.cpu(), ref["target_hidden_state"].detach()) < 3e-2
    assert rel(out.predictor_output.last_hidden_state.detach().cpu(), blob["out.predictor"]) < 3e-2      # the reference class itself
    for k, p in model.named_parameters():
        assert float((leaf[k].grad - blob["grad." + k]).abs().max()) <= 1e-4 * float(blob["grad." + k].abs().max()) + 1e-8, k  # oracle == reference
        r = leaf[k].grad
        if r is None or float(r.norm()) < 1e-10:
            continue
        e = float((p.grad.cpu() - r).norm() / r.norm())
        assert e < (2e-1 if k.endswith("attention.key.bias") else 8e-2), (k, e)
    # a single pair handed over as 1-element lists is the old path; mismatching list lengths / token counts raise
    with pytest.raises(ValueError):
        model(pixel_values_videos=pix.cuda(), context_mask=[cms[0].cuda()], target_mask=[t.cuda() for t in tms])
    with pytest.raises(ValueError):
        model(pixel_values_videos=pix.cuda(), context_mask=[cms[0].cuda(), cms[1][:, :10].cuda()], target_mask=[t.cuda() for t in tms])


@pytest.mark.parametrize("B", [1, 2])
def test_vjepa_restricted_last_encoder_block_equals_the_full_forward(B, monkeypatch):
    """smbv_needed / smbv_vjepa_cfg.enc_needed: with the encoder's tokens permuted to [consumed | others] (RoPE on explicit ids) the
    last encoder block computes the consumed rows only.  Loss, predictor output, the target encoder's target rows and EVERY gradient must equal the
    unrestricted step's up to summation order (the permutation reorders keys and tokens: same products, other order of the fp32 sums - bf16 roundings
    flip on single elements): outputs 2e-3 of scale, gradients ||d|| <= 1e-2 ||g||."""
    from smb_vision_amd.modeling_vjepa import make_target_encoder, vjepa_loss
    blob, c = load("vjepa_tiny")
    p_on = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    p_tg = {k[7:]: v for k, v in blob.items() if k.startswith("tparam.")}
    model = build(c, p_on)
    target = make_target_encoder(build(c, p_tg))
    g = torch.Generator().manual_seed(3)
    pix = torch.rand(B, c["frames_per_clip"], 1, c["crop_size"], c["crop_size"], generator=g).cuda()
    n = (c["frames_per_clip"] // c["tubelet_size"]) * (c["crop_size"] // c["patch_size"]) ** 2
    perm = torch.randperm(n, generator=g)
    cm = perm[: n // 3][None].repeat(B, 1).cuda()                    # NOT sorted: the mask order must not matter
    tm = perm[n // 3:].sort().values[None].repeat(B, 1).cuda()
    inputs = {"pixel_values_videos": pix, "context_mask": [cm], "target_mask": [tm]}

    def step(full):
        monkeypatch.setenv("SMBV_VJEPA_FULL_ENCODER", "1" if full else "0")
        model.zero_grad(set_to_none=True)
        loss, out = vjepa_loss(model, target, None, inputs)
        loss.backward()
        torch.cuda.synchronize()
        with torch.no_grad():
            tgt = target(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm], skip_predictor=True, smbv_needed=None if full else "target").target_hidden_state
        return float(loss), out.predictor_output.last_hidden_state.detach().clone(), out.masked_hidden_state.clone(), tgt.clone(), \
            {k: p.grad.detach().clone() for k, p in model.named_parameters()}, out

    l0, p0, m0, t0, g0, o0 = step(True)
    l1, p1, m1, t1, g1, o1 = step(False)
    assert o0.last_hidden_state is not None and o1.last_hidden_state is None
    assert abs(l0 - l1) <= 1e-4 * abs(l0), (l0, l1)
    assert rel(p1, p0) < 2e-3 and rel(m1, m0) < 2e-3 and rel(t1, t0) < 2e-3, (rel(p1, p0), rel(m1, m0), rel(t1, t0))
    worst = ("", 0.0)
    for k in g0:
        nrm = float(g0[k].norm())
        if nrm < 1e-10:
            assert float(g1[k].norm()) < 1e-6, k
            continue
        r = float((g1[k] - g0[k]).norm()) / nrm
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r < (5e-2 if k.endswith("attention.key.bias") else 1e-2), (k, r)
    print(f"restricted last encoder block, B={B}: loss {l1:.6f} vs {l0:.6f}, worst gradient difference {worst}")


@pytest.mark.parametrize("skip_predictor", [False, True])
def test_vjepa_gradients_through_the_encoder_outputs_match_the_oracle(skip_predictor):
    """VERDICT r2 'missing' item 4: VJEPA2Model.forward returns differentiable last_hidden_state / masked_hidden_state / target_hidden_state
    (modeling_vjepa.py:1137-1163); a loss that reads them (and, optionally, the predictor output) must give the oracle's gradients for every parameter."""
    import oracle.vjepa_oracle as vo
    c = dict(TINY)
    cfg = vo.VJEPAConfig(**c)
    params = vo.init_params(cfg, seed=9, nonzero_small=True)
    n = (c["frames_per_clip"] // c["tubelet_size"]) * (c["crop_size"] // c["patch_size"]) ** 2
    g = torch.Generator().manual_seed(13)
    perm = torch.randperm(n, generator=g)
    cm, tm = perm[: n // 2].sort().values[None].repeat(2, 1), perm[n // 2:].sort().values[None].repeat(2, 1)
    pix = torch.rand(2, c["frames_per_clip"], 1, c["crop_size"], c["crop_size"], generator=g)
    d = c["hidden_size"]
    w_last, w_m, w_t, w_p = (torch.randn(2, k, d, generator=g) for k in (n, cm.shape[1], tm.shape[1], tm.shape[1]))

    def functional(o, pred):
        v = (o["last"] * w_last.to(o["last"].device)).sum() + 0.5 * (o["masked"] * w_m.to(o["last"].device)).sum() + 0.25 * (o["target"] * w_t.to(o["last"].device)).sum()
        return v if pred is None else v + (pred * w_p.to(pred.device)).sum()

    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    ro = vo.model_forward(leaf, cfg, pix, [cm], [tm], skip_predictor=skip_predictor)
    functional(dict(last=ro["last_hidden_state"], masked=ro["masked_hidden_state"], target=ro["target_hidden_state"]), ro["predictor_output"]).backward()
    model = build(c, params)
    out = model(pixel_values_videos=pix.cuda(), context_mask=[cm.cuda()], target_mask=[tm.cuda()], skip_predictor=skip_predictor)
    assert (out.predictor_output is None) == skip_predictor
    functional(dict(last=out.last_hidden_state, masked=out.masked_hidden_state, target=out.target_hidden_state),
               None if skip_predictor else out.predictor_output.last_hidden_state).backward()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for k, p_ in model.named_parameters():
        ref = leaf[k].grad
        if ref is None or float(ref.norm()) < 1e-10:
            assert p_.grad is None or float(p_.grad.norm()) < 1e-6, k
            continue
        r = float((p_.grad.cpu() - ref).norm() / ref.norm())
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r < (2e-1 if k.endswith("attention.key.bias") else 6e-2), (k, r)
    print(f"gradients through the encoder outputs (skip_predictor={skip_predictor}): worst {worst}")


def test_host_built_index_arrays_give_the_same_step():
    """run_vjepa.collate_fn ships the step's index arrays (modeling_vjepa.index_plumbing, built on the host); the forward that receives them must be
    bit-identical to the one that derives them on the device, and arrays built for another restriction must be ignored, not misused."""
    from smb_vision_amd.modeling_vjepa import index_plumbing, make_target_encoder, vjepa_loss
    from smb_vision_amd.run_vjepa import collate_fn
    blob, c = load("vjepa_tiny")
    p_on = {k[6:]: v for k, v in blob.items() if k.startswith("param.")}
    model = build(c, p_on)
    target = make_target_encoder(build(c, p_on))
    pix = blob["in.pixel_values_videos"]
    cm, tm = blob["in.context_mask"], blob["in.target_mask"]
    examples = [{"image": pix[b], "context_mask": cm[0], "target_mask": tm[0]} for b in range(pix.shape[0])]
    import random
    random.seed(0)
    batch = collate_fn(examples, num_patches=model.num_patches)
    assert "smbv_plumbing_context" in batch and "smbv_plumbing_target" in batch
    dev = lambda x: x.cuda() if torch.is_tensor(x) else ({k: dev(v) for k, v in x.items()} if isinstance(x, dict) else ([dev(v) for v in x] if isinstance(x, list) else x))
    with_arrays = dev(batch)
    without = {k: v for k, v in with_arrays.items() if not k.startswith("smbv_")}
    res = []
    for inputs in (with_arrays, without):
        model.zero_grad(set_to_none=True)
        loss, out = vjepa_loss(model, target, None, inputs)
        loss.backward()
        res.append((loss.detach().clone(), out.predictor_output.last_hidden_state.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))
    # arrays of the wrong restriction: ignored (the unrestricted forward derives its own)
    wrong = index_plumbing([with_arrays["context_mask"][0].long()], [with_arrays["target_mask"][0].long()], model.num_patches, "target")
    a = model(pixel_values_videos=with_arrays["pixel_values_videos"], context_mask=with_arrays["context_mask"], target_mask=with_arrays["target_mask"],
              smbv_plumbing=wrong)
    b = model(pixel_values_videos=with_arrays["pixel_values_videos"], context_mask=with_arrays["context_mask"], target_mask=with_arrays["target_mask"])
    assert torch.equal(a.last_hidden_state, b.last_hidden_state) and torch.equal(a.predictor_output.last_hidden_state, b.predictor_output.last_hidden_state)
    # arrays of the right shapes built from OTHER masks (a batch dict reused after its masks were changed): detected - compared on the device, read back
    # without a per-step synchronisation, raised at the next forward or by check_plumbing()
    model.check_plumbing()                                   # everything queued so far was consistent
    cmask = with_arrays["context_mask"][0].clone()
    tmask = with_arrays["target_mask"][0].clone()
    cmask[:, 0], tmask[:, 0] = with_arrays["target_mask"][0][:, 0], with_arrays["context_mask"][0][:, 0]      # same counts, one token swapped between the sets
    model(pixel_values_videos=with_arrays["pixel_values_videos"], context_mask=[cmask], target_mask=[tmask], smbv_needed="context",
          smbv_plumbing=with_arrays["smbv_plumbing_context"])
    with pytest.raises(ValueError, match="was not built from the context / target masks"):
        model.check_plumbing()
    model.check_plumbing()                                   # reported once
