"""GPU: gradient accumulation semantics and the data-parallel drop-in (SmbTrainer).

* Accumulation (scripts/training/run_vjepa.sh:28 trains with --gradient_accumulation_steps 4): the step's gradients are views of ONE persistent
  arena; two micro-batches must still sum, and ``zero_grad(set_to_none=False)`` must still work — for MIM and V-JEPA.
* DP: two processes on the one GPU over gloo (RCCL needs one device per rank; the bucket / event / averaging code is the same, the 8-GPU run is
  the driver's): ``p.grad`` of the real MIM model equals the mean of the two single-process gradients, and ``run_mim.main`` launched the way
  ``torch.distributed.run`` would (RANK / WORLD_SIZE / MASTER_* in the env) keeps the replicas identical through optimizer steps without a DDP wrap.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MIM_TINY = dict(image_size=64, num_frames=32, patch_size=16, tubelet_size=16, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                intermediate_size=256, decoder_hidden_size=64, decoder_num_hidden_layers=1, decoder_num_attention_heads=1, decoder_intermediate_size=128)
VJ_TINY = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
               mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)


def _mim_model(seed=0):
    from transformers import VideoMAEConfig
    from smb_vision_amd.modeling_videomae import SmbVideoMAEForPreTraining
    torch.manual_seed(seed)
    return SmbVideoMAEForPreTraining(VideoMAEConfig(num_channels=1, **MIM_TINY)).cuda().train()


def _mim_batch(seed):
    from smb_vision_amd.masking import MaskGenerator
    np.random.seed(seed)
    gen = MaskGenerator(64, 32, 16, 16, 0.6)
    pix = torch.rand(2, 32, 1, 64, 64, generator=torch.Generator().manual_seed(seed)).cuda()
    return pix, torch.stack([gen(), gen()]).cuda()


def _single_grads(model, fn):
    model.zero_grad(set_to_none=True)
    fn().backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("set_to_none", [True, False])
def test_mim_gradient_accumulation_sums_micro_batches(set_to_none):
    model = _mim_model()
    b1, b2 = _mim_batch(1), _mim_batch(2)
    f1 = lambda: model(pixel_values=b1[0], bool_masked_pos=b1[1]).loss * 0.5   # HF divides the loss by the accumulation steps
    f2 = lambda: model(pixel_values=b2[0], bool_masked_pos=b2[1]).loss * 0.5
    g1, g2 = _single_grads(model, f1), _single_grads(model, f2)
    # zero_grad(set_to_none=False) leaves p.grad ALIASING the arena, zeroed; the next backward must not lose what autograd adds to it
    model.zero_grad(set_to_none=set_to_none)
    f1().backward()
    f2().backward()
    torch.cuda.synchronize()
    for p, a, b in zip(model.parameters(), g1, g2):
        assert torch.allclose(p.grad, a + b, rtol=1e-5, atol=1e-9)
    # and a third round after zero_grad(set_to_none=False): p.grad now are private tensors, the arena views are added into them
    model.zero_grad(set_to_none=False)
    f2().backward()
    torch.cuda.synchronize()
    for p, b in zip(model.parameters(), g2):
        assert torch.allclose(p.grad, b, rtol=1e-5, atol=1e-9)


def test_vjepa_gradient_accumulation_sums_micro_batches():
    from transformers import VJEPA2Config
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    torch.manual_seed(0)
    model = SmbVJEPA2Model(VJEPA2Config(**VJ_TINY)).cuda().train()
    n = (VJ_TINY["frames_per_clip"] // VJ_TINY["tubelet_size"]) * (VJ_TINY["crop_size"] // VJ_TINY["patch_size"]) ** 2

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        pix = torch.rand(2, VJ_TINY["frames_per_clip"], 1, VJ_TINY["crop_size"], VJ_TINY["crop_size"], generator=g).cuda()
        perm = torch.randperm(n, generator=g)
        cm, tm = perm[: n // 2].sort().values, perm[n // 2:].sort().values
        return pix, cm[None].repeat(2, 1).cuda(), tm[None].repeat(2, 1).cuda()

    def fn(b):
        return lambda: model(pixel_values_videos=b[0], context_mask=[b[1]], target_mask=[b[2]]).predictor_output.last_hidden_state.square().mean()

    b1, b2 = batch(1), batch(2)
    g1, g2 = _single_grads(model, fn(b1)), _single_grads(model, fn(b2))
    for stn in (True, False):
        model.zero_grad(set_to_none=stn)
        fn(b1)().backward()
        fn(b2)().backward()
        torch.cuda.synchronize()
        for p, a, b in zip(model.parameters(), g1, g2):
            assert torch.allclose(p.grad, a + b, rtol=1e-5, atol=1e-9)


def test_trainer_with_gradient_accumulation_matches_manual_loop(tmp_path):
    """SmbTrainer (HF Trainer + fused clip/AdamW) with --gradient_accumulation_steps 2 lands on the same weights as a hand-written loop with
    torch's clip_grad_norm_ + AdamW over the same four micro-batches."""
    from transformers import TrainingArguments
    from smb_vision_amd.run_mim import collate_fn
    from smb_vision_amd.trainer import SmbTrainer
    batches = [_mim_batch(s) for s in (1, 2, 3, 4)]
    data = [{"image": pix[i].cpu(), "mask": m[i].cpu()} for pix, m in batches for i in range(2)]
    args = TrainingArguments(output_dir=str(tmp_path), per_device_train_batch_size=2, gradient_accumulation_steps=2, max_steps=2, learning_rate=1e-3,
                             lr_scheduler_type="constant", weight_decay=0.05, max_grad_norm=0.5, report_to="none", save_strategy="no", bf16=True,
                             dataloader_num_workers=0, remove_unused_columns=False, logging_steps=1, seed=0, adam_beta2=0.95)
    model = _mim_model(3)
    start = {k: v.detach().clone() for k, v in model.state_dict().items()}

    class InOrder(SmbTrainer):
        def _get_train_sampler(self, *a, **k):
            return torch.utils.data.SequentialSampler(self.train_dataset)

    tr = InOrder(model=model, args=args, train_dataset=data, data_collator=collate_fn)
    tr.train()
    from smb_vision_amd.optim import SmbvAdamW
    assert isinstance(getattr(tr.optimizer, "optimizer", tr.optimizer), SmbvAdamW)
    assert any("grad_norm" in h for h in tr.state.log_history)

    ref = _mim_model(3)
    ref.load_state_dict(start)
    decay = set(tr.get_decay_parameter_names(ref))
    opt = torch.optim.AdamW([{"params": [p for n, p in ref.named_parameters() if n in decay], "weight_decay": 0.05},
                             {"params": [p for n, p in ref.named_parameters() if n not in decay], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    for s in range(2):
        opt.zero_grad(set_to_none=True)
        for pix, m in batches[2 * s: 2 * s + 2]:
            (ref(pixel_values=pix, bool_masked_pos=m).loss / 2).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
    # Adam normalises the update, and a 1-ulp difference after step 1 (fused vs torch optimizer arithmetic) flips bf16 roundings in step 2's
    # forward: compare the two-step UPDATES, per tensor, at 1e-2 of their size (a lost or doubled micro-batch changes them by O(1))
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        upd = float((b - start[k]).norm())
        assert upd > 0, k
        assert float((a - b).norm()) <= 1e-2 * upd, (k, float((a - b).norm()), upd)


def test_stock_transformers_trainer_trains_the_native_model_as_integration_promises(tmp_path):
    """"keep Trainer(...) unchanged, swap the model class" - the UNMODIFIED ``transformers.Trainer`` (accelerate's bf16 autocast wrap and
    backward, its ``clip_grad_norm_`` on the arena-view gradients, torch's own AdamW, ``zero_grad``) drives ``SmbVideoMAEForPreTraining`` for three steps and
    lands where ``SmbTrainer`` (fused clip + AdamW, copy-stream uploads) lands from the same start on the same batches.  Tolerance: per tensor, the two
    runs' three-step UPDATES differ by <= 1e-2 of the update (fused vs torch optimizer arithmetic differs by an ulp after step 1; a dropped clip, a missing
    decay group or a stale gradient moves it by O(1))."""
    from transformers import Trainer, TrainingArguments
    from smb_vision_amd.run_mim import collate_fn
    from smb_vision_amd.trainer import SmbTrainer
    batches = [_mim_batch(s) for s in (1, 2, 3)]
    data = [{"image": pix[i].cpu(), "mask": m[i].cpu()} for pix, m in batches for i in range(2)]

    def run(cls, sub):
        args = TrainingArguments(output_dir=str(tmp_path / sub), per_device_train_batch_size=2, max_steps=3, learning_rate=1e-3, lr_scheduler_type="constant",
                                 weight_decay=0.05, max_grad_norm=1.0, report_to="none", save_strategy="no", bf16=True, dataloader_num_workers=0,
                                 remove_unused_columns=False, logging_steps=1, seed=0, adam_beta2=0.95)
        model = _mim_model(3)
        start = {k: v.detach().clone() for k, v in model.state_dict().items()}

        class InOrder(cls):
            def _get_train_sampler(self, *a, **k):
                return torch.utils.data.SequentialSampler(self.train_dataset)

            def get_decay_parameter_names(self, m):   # the same decay split for both (SmbTrainer's: HF's own split of the reference classes, golden/decay_names.json)
                return SmbTrainer.get_decay_parameter_names(self, m)

        tr = InOrder(model=model, args=args, train_dataset=data, data_collator=collate_fn)
        tr.train()
        torch.cuda.synchronize()
        losses = [h["loss"] for h in tr.state.log_history if "loss" in h]
        norms = [h["grad_norm"] for h in tr.state.log_history if "grad_norm" in h]
        return start, {k: v.detach().clone() for k, v in model.state_dict().items()}, losses, norms, tr

    s0, stock, l_stock, n_stock, tr_stock = run(Trainer, "stock")
    s1, ours, l_ours, n_ours, tr_ours = run(SmbTrainer, "smb")
    from smb_vision_amd.optim import SmbvAdamW
    assert not isinstance(getattr(tr_stock.optimizer, "optimizer", tr_stock.optimizer), SmbvAdamW)      # torch's optimizer really ran
    assert isinstance(getattr(tr_ours.optimizer, "optimizer", tr_ours.optimizer), SmbvAdamW)
    assert len(l_stock) == 3 and l_stock[0] == pytest.approx(l_ours[0], rel=1e-6) and l_stock[2] == pytest.approx(l_ours[2], rel=2e-3)
    assert n_stock[0] == pytest.approx(n_ours[0], rel=1e-4)                                              # the same pre-clip global norm is logged
    for k in s0:
        assert torch.equal(s0[k], s1[k])
        upd = float((stock[k] - s0[k]).norm())
        assert upd > 0, k
        assert float((stock[k] - ours[k]).norm()) <= 1e-2 * upd, (k, float((stock[k] - ours[k]).norm()), upd)


def test_run_mim_trains_from_a_volume_cache_directory(tmp_path):
    """``--volume_cache_dir`` (the role of the reference's --json_path / --cache_dir, src/run_mim.py:61, 140-143, 279-290): four steps over eight raw int16
    volumes written by ``volcache`` (ragged shapes: padded and cropped on the device), read into the feeder's page-locked ring and uploaded through
    ``VolumeFeeder``; the losses and the final weights equal - bit for bit - a run that is handed the SAME volumes as fp32 ``image`` tensors in memory
    (prepared by the feed oracle) with the same masks.  Then an evaluation-only invocation over the same directory (raw batches converted inline)."""
    from safetensors.torch import load_file
    import oracle.feed_oracle as fo
    from smb_vision_amd import run_mim, volcache as vc
    cache = tmp_path / "cache"
    os.makedirs(cache)
    rng = np.random.default_rng(11)
    raws = []
    for i in range(8):
        r = rng.integers(-1400, 1400, size=(56 + 5 * (i % 3), 70 - 4 * (i % 2), 28 + 3 * (i % 4)), dtype=np.int16)
        vc.write_volume(str(cache / f"vol{i:02d}.smbvol"), r)
        raws.append(r)
    tiny = ["--image_size", "64", "--depth", "32", "--patch_size", "16", "--mask_patch_size", "16", "--mask_ratio", "0.6", "--per_device_train_batch_size", "2",
            "--report_to", "none", "--bf16", "true", "--hidden_size", "128", "--num_hidden_layers", "2", "--num_attention_heads", "2", "--intermediate_size", "256",
            "--decoder_hidden_size", "64", "--decoder_num_hidden_layers", "1", "--decoder_num_attention_heads", "1", "--decoder_intermediate_size", "128",
            "--learning_rate", "1e-3", "--logging_steps", "1", "--max_grad_norm", "1.0", "--dataloader_num_workers", "0", "--save_strategy", "no", "--max_steps", "4",
            "--seed", "3"]
    tr_a = run_mim.build_trainer(["--output_dir", str(tmp_path / "a"), "--volume_cache_dir", str(cache), *tiny])
    assert getattr(tr_a.train_dataset, "smbv_raw_volumes", False) and len(tr_a.train_dataset) == 8
    from smb_vision_amd.trainer import VolumeFeedLoader
    assert isinstance(tr_a.get_train_dataloader(), VolumeFeedLoader)
    masks = [tr_a.train_dataset[i]["mask"] for i in range(8)]
    tr_a.train()
    tr_a.save_model()

    tr_b = run_mim.build_trainer(["--output_dir", str(tmp_path / "b"), *tiny])
    tr_b.train_dataset = [{"image": torch.from_numpy(fo.prepare_volume(raws[i], 64, 32)), "mask": masks[i]} for i in range(8)]
    tr_b.train()
    tr_b.save_model()
    la = [h["loss"] for h in tr_a.state.log_history if "loss" in h]
    lb = [h["loss"] for h in tr_b.state.log_history if "loss" in h]
    assert len(la) == 4 and la == lb, (la, lb)
    wa, wb = load_file(str(tmp_path / "a" / "model.safetensors")), load_file(str(tmp_path / "b" / "model.safetensors"))
    assert all(torch.equal(wa[k], wb[k]) for k in wa)

    # evaluation only: nothing trains, nothing is overwritten, the raw batches of the evaluation loader are converted inline
    before = os.path.getmtime(tmp_path / "a" / "model.safetensors")
    m = run_mim.main(["--output_dir", str(tmp_path / "a"), "--do_eval", "true", "--eval_volume_cache_dir", str(cache), "--model_name_or_path", str(tmp_path / "a"),
                      "--per_device_eval_batch_size", "2", *tiny])
    assert "train_loss" not in m and 0.0 < m["eval_loss"] < 2.0 and os.path.getmtime(tmp_path / "a" / "model.safetensors") == before


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q, overlap=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from smb_vision_amd.dp import DataParallelEngine
        torch.cuda.set_device(0)
        model = _mim_model(100 + rank)                   # different initial weights per rank: the broadcast must fix that
        # gloo: the engine's SUM + scale path (RCCL uses AVG).  overlap=True still drives everything else the RCCL launch does: per-group HIP events
        # created here, recorded by smbv_mim_backward as each gradient group completes, waited for by the communication stream bucket by bucket
        dp = DataParallelEngine(model, overlap=overlap, bucket_bytes=1 << 16)
        dp.broadcast_parameters(model)
        pix, mask = _mim_batch(10 + rank)                # different data per rank
        model(pixel_values=pix, bool_masked_pos=mask).loss.backward()
        torch.cuda.synchronize()
        # numpy arrays are pickled by value (torch tensors would travel as shared-memory handles that die with this process)
        q.put((rank, {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_mim_dp_param_grads_are_the_mean_of_the_rank_gradients(overlap):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
    assert all(p.exitcode == 0 for p in ps)
    (_, ga, wa), (_, gb, wb) = [(r, {k: torch.from_numpy(v) for k, v in g.items()}, {k: torch.from_numpy(v) for k, v in w.items()}) for r, g, w in res]
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k               # rank 0's parameters everywhere
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k               # both ranks hold the same averaged p.grad
    # single-process gradients of the same two batches with rank 0's weights
    model = _mim_model(100)
    model.load_state_dict(wa)
    singles = []
    for r in range(2):
        pix, mask = _mim_batch(10 + r)
        singles.append(_single_grads(model, lambda: model(pixel_values=pix, bool_masked_pos=mask).loss))
    for (k, _), a, b in zip(model.named_parameters(), *singles):
        assert torch.allclose(ga[k].cuda(), 0.5 * (a + b), rtol=1e-5, atol=1e-9), k


def test_run_mim_under_a_two_process_launch_keeps_replicas_in_sync(tmp_path):
    """``smb_vision_amd.run_mim`` (its ``build_trainer`` + ``train``, via tests/dp_worker_mim.py) as torch.distributed.run starts it (2 ranks, gloo, both on the one GPU): no DDP wrap, the engine
    averages, SmbvAdamW steps; after 3 optimizer steps both ranks hold bit-identical weights that differ from the start."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_worker_mim.py"), str(tmp_path / f"state{rank}.pt"), "--output_dir", str(tmp_path / f"out{rank}"), "--image_size", "64", "--depth", "32",
               "--patch_size", "16", "--mask_patch_size", "16", "--mask_ratio", "0.6", "--max_steps", "3", "--per_device_train_batch_size", "2",
               "--report_to", "none", "--bf16", "true", "--hidden_size", "128", "--num_hidden_layers", "2", "--num_attention_heads", "2",
               "--intermediate_size", "256", "--decoder_hidden_size", "64", "--decoder_num_hidden_layers", "1", "--decoder_num_attention_heads", "1",
               "--decoder_intermediate_size", "128", "--learning_rate", "1e-3", "--logging_steps", "1", "--save_strategy", "no", "--max_grad_norm", "1.0",
               "--dataloader_num_workers", "0", "--ddp_backend", "gloo", "--seed", "5"]
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    s0 = torch.load(tmp_path / "state0.pt", weights_only=True)
    s1 = torch.load(tmp_path / "state1.pt", weights_only=True)
    assert s0["meta"]["wrapped"] is False and s0["meta"]["dp_world"] == 2 and s0["meta"]["optimizer"] == "SmbvAdamW"
    assert s0["meta"]["first_batch_sum"] != s1["meta"]["first_batch_sum"]          # the ranks really saw different shards
    moved = 0
    for k in s0["final"]:
        assert torch.equal(s0["final"][k], s1["final"][k]), k
        assert torch.equal(s0["initial"][k], s1["initial"][k]), k
        moved += int(not torch.equal(s0["final"][k], s0["initial"][k]))
    assert moved > 0.9 * len(s0["final"])


def test_bench_launches_its_own_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2 ...` exactly as the driver starts it - no torch.distributed.run in front, WORLD_SIZE unset: the parent (which makes no GPU
    call) starts the two ranks as children, relays rank 0's ONE JSON line and returns their exit code.  gloo, both ranks on the one GPU (RCCL needs one
    device per rank; the 8-GPU curve is the driver's)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-extra-configs", "--no-trainer", "--no-feed"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["replicas_in_sync"] is True and line["config"]["parallelism"] == "dp2"
    assert line["steps"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["roofline"]["frac"] > 0
    # the all-reduce plan travels in the line: 25-MB buckets - every ViT-base encoder block on its own, 388.7 MB in all (dp.py)
    plan = line["config"]["dp"]
    assert plan["grad_dtype"] == "fp32" and len(plan["bucket_MB"]) == 14 and abs(sum(plan["bucket_MB"]) - 388.7) < 1.5 and plan["bucket_MB"][-1] < 13
    # a rank that fails must fail the launch: an unknown backend makes every rank raise in init_process_group
    bad = subprocess.run([a if a != "gloo" else "no-such-backend" for a in cmd], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and not bad.stdout.strip()


def _rccl_world1_worker(port, q, which):
    """One rank, backend "nccl" (= RCCL): the engine's overlap path with the collectives forced on at world size 1 - per-group HIP events recorded by
    smbv_*_backward, the communication stream waiting on them bucket by bucket, ``all_reduce(AVG, async_op=True)`` under that stream, the compute
    stream joined before the optimizer - on the real device."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from smb_vision_amd.dp import DataParallelEngine
        from smb_vision_amd.optim import SmbvAdamW
        grad_dtype = "bf16" if which.endswith("-bf16") else "fp32"
        if which.startswith("mim"):
            model = _mim_model(100)
            pix, mask = _mim_batch(10)
            fn = lambda: model(pixel_values=pix, bool_masked_pos=mask).loss
        else:
            from transformers import VJEPA2Config
            from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
            torch.manual_seed(0)
            model = SmbVJEPA2Model(VJEPA2Config(**VJ_TINY)).cuda().train()
            n = (VJ_TINY["frames_per_clip"] // VJ_TINY["tubelet_size"]) * (VJ_TINY["crop_size"] // VJ_TINY["patch_size"]) ** 2
            g = torch.Generator().manual_seed(1)
            pix = torch.rand(2, VJ_TINY["frames_per_clip"], 1, VJ_TINY["crop_size"], VJ_TINY["crop_size"], generator=g).cuda()
            perm = torch.randperm(n, generator=g)
            cm, tm = perm[: n // 2].sort().values[None].repeat(2, 1).cuda(), perm[n // 2:].sort().values[None].repeat(2, 1).cuda()
            fn = lambda: model(pixel_values_videos=pix, context_mask=[cm], target_mask=[tm]).predictor_output.last_hidden_state.square().mean()
        plain = _single_grads(model, fn)                                  # no engine attached
        dp = DataParallelEngine(model, overlap=True, bucket_bytes=1 << 16, force_collectives=True, grad_dtype=grad_dtype)
        dp.broadcast_parameters(model)                                     # RCCL broadcast (identity at world 1)
        through = _single_grads(model, fn)                                 # events -> comm stream -> all_reduce(AVG, async) -> join
        n_buckets = len(dp._buckets)
        if grad_dtype == "bf16":   # cast -> AVG over one rank -> cast back: exactly the bf16 rounding of the fp32 gradients, still fp32 tensors
            same = all(torch.equal(a.bfloat16().float(), b) and b.dtype == torch.float32 for a, b in zip(plain, through))
        else:
            same = all(torch.equal(a, b) for a, b in zip(plain, through))
        opt = SmbvAdamW(model.parameters(), lr=1e-3, max_grad_norm=1.0)
        before = [p.detach().clone() for p in model.parameters()]
        for _ in range(3):                                                 # three full steps: backward + collectives + optimizer, no deadlock
            opt.zero_grad(set_to_none=True)
            fn().backward()
            opt.step()
        torch.cuda.synchronize()
        moved = sum(int(not torch.equal(a, p)) for a, p in zip(before, model.parameters()))
        q.put(dict(same=same, n_buckets=n_buckets, overlap=dp.comm_stream is not None, n_events=len(dp._events), moved=moved, n_params=len(before),
                   backend=str(dist.get_backend())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["mim", "vjepa", "mim-bf16"])
def test_rccl_collective_path_runs_on_hardware_at_world_size_one(which):
    """the RCCL lines of dp.py (ReduceOp.AVG with async_op under the side stream, the event-gated bucket loop) execute on an MI355X:
    backend "nccl" at world size 1 with DataParallelEngine(force_collectives=True).  An averaging all-reduce over one rank is the identity, so the
    gradients must come out bit for bit as without the engine, and optimizer steps behind it must not deadlock.  (The 8-GPU curve stays the driver's.)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q, which))
    p.start()
    try:
        res = q.get(timeout=420)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    assert res["backend"] == "nccl" and res["overlap"] and res["n_events"] >= 4 and res["n_buckets"] >= 3, res
    assert res["same"], "gradients changed through the world-size-1 averaging all-reduce"
    assert res["moved"] > 0.9 * res["n_params"], res


def test_accumulation_keeps_gradient_addresses_and_the_optimizer_table_stable():
    """under gradient accumulation the gradients a step hands to the optimizer live in ONE persistent accumulation arena (views at fixed
    offsets), so SmbvAdamW builds its device-side chunk table once instead of every step."""
    from smb_vision_amd.optim import SmbvAdamW
    model = _mim_model(5)
    opt = SmbvAdamW(model.parameters(), lr=1e-3, max_grad_norm=1.0)
    builds = []
    stock = opt._build
    opt._build = lambda items: (builds.append(1), stock(items))[1]
    ptrs = []
    for step in range(3):
        for mb in range(2):
            pix, mask = _mim_batch(20 + 2 * step + mb)
            (model(pixel_values=pix, bool_masked_pos=mask).loss / 2).backward()
        ptrs.append([p.grad.data_ptr() for p in model.parameters()])
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    assert ptrs[0] == ptrs[1] == ptrs[2]
    assert len(builds) == 1, builds
    lo = model._gacc.data_ptr()
    assert all(lo <= a < lo + model._gacc.numel() * 4 for a in ptrs[0])
