"""CPU: the data-parallel engine's bucketing and averaging with world_size-2 gloo processes (the N>1 path of bench.py;
on the GPU the same code runs over RCCL with HIP events gating each bucket on a side stream)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smb_vision_amd.dp import DataParallelEngine, allreduce_mean_, merge_ranges


def test_merge_ranges_covers_everything_in_completion_order():
    # completion order of a 2-layer encoder / 1-layer decoder model: head, dec0, e2d, enc1, enc0, patch-embed (arena is in forward order)
    ranges = [(900, 1000), (700, 900), (600, 700), (300, 600), (100, 300), (0, 100)]
    b = merge_ranges(ranges, 250)
    assert b == [(700, 1000, 1), (300, 700, 3), (0, 300, 5)]
    covered = sorted((a, e) for a, e, _ in b)
    assert covered[0][0] == 0 and covered[-1][1] == 1000 and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    # every bucket is gated on the LAST group that completes inside it
    for a, e, last in b:
        assert ranges[last][0] == a or ranges[last][1] == e
    assert merge_ranges(ranges, 1) == [(a, e, i) for i, (a, e) in enumerate(ranges)]
    assert merge_ranges(ranges, 10 ** 9) == [(0, 1000, 5)]


class FakeModel:
    """Stands in for SmbVideoMAEForPreTraining on the CPU: a flat fp32 gradient arena + completion-order group ranges."""

    def __init__(self, n, ranges):
        self._g = torch.zeros(n)
        self._r = ranges
        self.dp = None

    def flat_grads(self):
        return self._g

    def grad_group_ranges(self):
        return self._r

    def parameters(self):
        return [self._g]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        ranges = [(900, 1000), (700, 900), (600, 700), (300, 600), (100, 300), (0, 100)]
        model = FakeModel(n, ranges)
        eng = DataParallelEngine(model, bucket_bytes=4 * 250)
        assert model.dp is eng
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(n, generator=g)
        model.flat_grads().copy_(local)
        eng.allreduce_after_backward(model)
        # reference: plain mean over ranks of the whole arena
        ref = local.clone()
        allreduce_mean_(ref)
        q.put((rank, bool(torch.allclose(model.flat_grads(), ref, rtol=0, atol=1e-7)), model.flat_grads().sum().item()))
        # parameter broadcast makes replicas identical
        p = torch.full((n,), float(rank))
        model._g = p
        eng.broadcast_parameters(model, src=0)
        q.put((rank, bool((p == 0).all()), 0.0))
        # ... and travels coalesced: many tensors of mixed shapes in few collectives (one flat buffer per <= chunk_bytes), values and order preserved
        many = [torch.full((3 + i % 5, 2 + i % 3), float(rank * 100 + i)) for i in range(40)]

        class Many:
            def parameters(self):
                return many
        eng.broadcast_parameters(Many(), src=0)
        ok = all(bool((t == float(i)).all()) for i, t in enumerate(many)) and eng.broadcast_collectives == 1
        eng.broadcast_parameters(Many(), src=0, chunk_bytes=400)     # forced into several chunks
        ok = ok and all(bool((t == float(i)).all()) for i, t in enumerate(many)) and 1 < eng.broadcast_collectives < 40
        q.put((rank, ok, 0.0))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(6)]
    assert all(ok for _, ok, _ in res)
    sums = [s for r, ok, s in res if s != 0.0]
    assert len(sums) == 2 and abs(sums[0] - sums[1]) < 1e-5  # both ranks hold the same averaged gradients


def _worker_bf16(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        ranges = [(900, 1000), (700, 900), (600, 700), (300, 600), (100, 300), (0, 100)]
        model = FakeModel(n, ranges)
        eng = DataParallelEngine(model, bucket_bytes=4 * 250, grad_dtype="bf16")
        local = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank)) * (10.0 ** torch.randint(-6, 3, (n,), generator=torch.Generator().manual_seed(7)))
        model.flat_grads().copy_(local)
        eng.allreduce_after_backward(model)
        got = model.flat_grads().clone()
        ref = local.clone()
        allreduce_mean_(ref)                      # the fp32 average
        both = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(both, got)
        identical = all(torch.equal(both[0], b) for b in both)
        # stated tolerance: each addend and the mean are rounded to bf16 (8 significant bits): |got - ref| <= 2^-8 (|g_0| + ... + |g_{w-1}|) / w + 2^-8 |ref|
        mags = local.abs().clone()
        allreduce_mean_(mags)
        bound = 2.0 ** -8 * (mags + ref.abs()) + 1e-30
        q.put((rank, identical, bool(((got - ref).abs() <= bound).all()), got.dtype == torch.float32,
               float(((got - ref).norm() / ref.norm()))))
    finally:
        dist.destroy_process_group()


def test_bf16_compressed_allreduce_world2_gloo_keeps_replicas_identical():
    """grad_dtype="bf16" (the reference stack's DDP bf16_compress_hook option): replicas bit-identical, every element within the two bf16
    roundings of the fp32 average, the arena itself stays fp32."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(2)]
    for _, identical, within, fp32, rel in res:
        assert identical and within and fp32
        assert rel <= 2.0 ** -7


def test_grad_dtype_is_validated():
    with pytest.raises((RuntimeError, ValueError)):
        DataParallelEngine(FakeModel(4, [(0, 4)]), grad_dtype="fp16")


def test_engine_requires_process_group():
    with pytest.raises(RuntimeError):
        DataParallelEngine(FakeModel(4, [(0, 4)]))


def test_vjepa_gradient_groups_tile_the_arena():
    """SmbVJEPA2Model.grad_group_ranges(): the completion groups smbv_vjepa_backward signals (4 + layers + predictor layers events) are
    disjoint, cover the whole flat arena, and come in reverse forward order, so that adjacent groups merge into contiguous buckets."""
    from transformers import VJEPA2Config
    from smb_vision_amd.modeling_vjepa import SmbVJEPA2Model
    c = dict(patch_size=8, crop_size=32, frames_per_clip=8, tubelet_size=2, in_chans=1, hidden_size=128, num_hidden_layers=3, num_attention_heads=2,
             mlp_ratio=4.0, pred_hidden_size=64, pred_num_hidden_layers=2, pred_num_attention_heads=2, pred_num_mask_tokens=10, pred_mlp_ratio=4.0)
    m = SmbVJEPA2Model(VJEPA2Config(**c))
    m._ensure_device_state(torch.device("cpu"))
    r = m.grad_group_ranges()
    assert len(r) == 4 + 3 + 2
    assert all(a < b for a, b in r)
    assert [b for _, b in r[1:]] == [a for a, _ in r[:-1]]          # each group ends where the previously completed one starts
    assert r[-1][0] == 0 and r[0][1] == m.flat_grads().numel()
    b = merge_ranges(r, 1 << 62)
    assert b == [(0, m.flat_grads().numel(), len(r) - 1)]
