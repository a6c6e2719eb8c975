"""Data-parallel gradient averaging for the MIM step: RCCL all-reduce over xGMI on a side HIP stream, overlapped with the
rest of the backward.  Replaces the accelerate/DDP wrap of the reference (``acc_configs/multi_gpu_config.yaml:3`` ->
``DistributedDataParallel`` with 25 MB buckets).

One process per GPU (``torch.distributed`` backend "nccl" == RCCL on ROCm).  The model's gradients live in ONE flat fp32
arena; ``smbv_mim_backward`` records a HIP event on the compute stream each time a contiguous group of that arena is final
(head, each decoder layer, encoder_to_decoder, each encoder layer, patch embedding — in that order).  The whole backward is
enqueued by one C call, after which this engine enqueues, per group, ``comm_stream.wait_event(ev)`` + an averaging
all-reduce of that slice — so on the GPU timeline each all-reduce starts the moment its gradients exist and runs beside the
remaining backward kernels.  The compute stream only waits for the communication stream once, before the optimizer.
Small adjacent groups are merged up to ``bucket_bytes`` (xGMI ring all-reduce is per-link bound; ≥16 MB per call keeps the
links busy).  The default is DDP's own 25 MB (what the reference runs with): a ViT-base encoder block's gradients are 28.4 MB, so every block
is a bucket of its own and its all-reduce starts when ITS backward ends - at 32 MB the blocks travelled in pairs, the first of a pair waiting
~0.75 ms for the second, and the exchange left exposed behind the LAST block was 57 + 13 MB instead of 28 + 13 MB.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def merge_ranges(ranges: Sequence[Tuple[int, int]], bucket_elems: int) -> List[Tuple[int, int, int]]:
    """Merge consecutive completion groups into buckets of at least ``bucket_elems`` elements when they are adjacent in the
    arena.  Returns [(start, end, index_of_last_group_in_bucket)] in completion order."""
    out: List[Tuple[int, int, int]] = []
    cur = None
    for i, (a, b) in enumerate(ranges):
        if cur is None:
            cur = [a, b, i]
        elif b == cur[0]:      # groups complete in reverse arena order, so the next one usually ends where the bucket starts
            cur[0], cur[2] = a, i
        elif a == cur[1]:
            cur[1], cur[2] = b, i
        else:
            out.append(tuple(cur))
            cur = [a, b, i]
        if cur[1] - cur[0] >= bucket_elems:
            out.append(tuple(cur))
            cur = None
    if cur is not None:
        out.append(tuple(cur))
    return out


def allreduce_mean_(flat: torch.Tensor, group=None, force: bool = False):
    """In-place mean over ranks (AVG where the backend has it, SUM + scale otherwise, e.g. gloo on CPU).  ``force``: issue the collective at world
    size 1 too (an identity; what lets a one-GPU box execute the RCCL code path, tests/test_accum_dp_gpu.py)."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return None
    if flat.is_cuda and "nccl" in str(dist.get_backend(group)).lower():  # RCCL: averaging collective, asynchronous on its own stream
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=True)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)  # gloo (CPU tensors, or device tensors when rehearsing on one GPU)
    flat.mul_(1.0 / world)
    return None


def allreduce_mean_bf16_(flat: torch.Tensor, group=None, force: bool = False):
    """The bf16-compressed exchange (DDP's ``bf16_compress_hook``): cast the fp32 slice to bf16, average THAT over the ranks, write the
    result back as fp32 - half the bytes on the links (194 MB instead of 389 MB per ViT-base step).  Every rank receives the same bf16 values (one reduction
    order per element in a ring / tree), so replicas stay bit-identical; against the fp32 average the result is off by the bf16 rounding of each rank's
    addend and of the mean (relative 2^-8 per element at worst, tests/test_dp_cpu.py).  Stream-ordered: call it under the stream that should carry the casts."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return
    buf = flat.to(torch.bfloat16)
    if flat.is_cuda and "nccl" in str(dist.get_backend(group)).lower():
        dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=True).wait()  # the CURRENT stream waits for RCCL's stream; the host does not
        flat.copy_(buf)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(buf.float().mul_(1.0 / world).to(torch.bfloat16))  # the mean rounded to bf16, as the AVG collective delivers it


class DataParallelEngine:
    def __init__(self, model, group=None, bucket_bytes: int = 25 << 20, overlap: bool = True, force_collectives: bool = False, grad_dtype: str = "fp32"):
        if not dist.is_initialized():
            raise RuntimeError("DataParallelEngine needs torch.distributed to be initialised (one process per GPU)")
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError(f"grad_dtype must be 'fp32' or 'bf16', got {grad_dtype!r}")
        self.grad_dtype = grad_dtype  # what travels over xGMI; the arena, the clip and AdamW stay fp32
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self.force_collectives = force_collectives  # run every bucket's collective even at world size 1 (the averaging all-reduce is then an identity)
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._events: List[torch.cuda.Event] = []
        self._ev_array = None
        self._buckets = None
        self.enqueue_lead: List[bool] = []
        self.broadcast_collectives = 0
        model.dp = self

    # -- called by the model ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, model, src: int = 0, chunk_bytes: int = 256 << 20):
        """Identical replicas at start (DDP broadcasts rank 0's parameters at wrap time).  The ~250 parameter tensors travel as a few flat buffers
        (<= ``chunk_bytes`` each: ONE collective for ViT-base's 389 MB at the default) instead of one broadcast per tensor - xGMI collectives are
        latency-bound below a few MB - grouped by dtype and device, order preserved."""
        from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
        params = [p.data for p in model.parameters()]
        self.broadcast_collectives = 0
        i = 0
        while i < len(params):
            j, nbytes = i, 0
            while j < len(params) and params[j].dtype == params[i].dtype and params[j].device == params[i].device and \
                    (j == i or nbytes + params[j].numel() * params[j].element_size() <= chunk_bytes):
                nbytes += params[j].numel() * params[j].element_size()
                j += 1
            group = params[i:j]
            if len(group) == 1:
                dist.broadcast(group[0], src=src, group=self.group)
            else:
                flat = _flatten_dense_tensors(group)
                dist.broadcast(flat, src=src, group=self.group)
                for p, v in zip(group, _unflatten_dense_tensors(flat, group)):
                    p.copy_(v)
            self.broadcast_collectives += 1
            i = j

    def _setup(self, model):
        ranges = model.grad_group_ranges()
        self._ranges = ranges
        self._buckets = merge_ranges(ranges, self.bucket_elems)
        if model.flat_grads().is_cuda and self.overlap:
            # highest priority: the casts / copies of the communication stream (and, with TORCH_NCCL_HIGH_PRIORITY=1 - bench.py and SmbTrainer set it before the
            # process group exists - RCCL's own kernels) are dispatched ahead of the backward's queued blocks instead of behind 0.5-2 ms attention launches
            self.comm_stream = torch.cuda.Stream(device=model.flat_grads().device, priority=-1)
            self._events = []
            for _ in ranges:
                ev = torch.cuda.Event()
                ev.record()  # forces creation of the underlying hipEvent_t so that its handle can cross the C ABI
                self._events.append(ev)
            arr = (C.c_void_p * len(ranges))()
            for i, ev in enumerate(self._events):
                arr[i] = ev.cuda_event
            self._ev_array = arr

    def event_array(self):
        if not self.overlap or self._ev_array is None:
            return None, 0
        return self._ev_array, len(self._events)

    def prepare(self, model):
        if self._buckets is None:
            self._setup(model)

    def allreduce_after_backward(self, model):
        flat = model.flat_grads()
        if self._buckets is None:
            self._setup(model)
        if self.world == 1 and not self.force_collectives:
            return
        force = self.force_collectives
        if self.comm_stream is None:  # no overlap (or CPU tensors in the gloo tests): plain bucketed all-reduce
            if self.grad_dtype == "bf16":
                for a, b, _ in self._buckets:
                    allreduce_mean_bf16_(flat[a:b], self.group, force)
                return
            works = [allreduce_mean_(flat[a:b], self.group, force) for a, b, _ in self._buckets]
            for w in works:
                if w is not None:
                    w.wait()
            return
        main = torch.cuda.current_stream()
        works = []
        # The whole backward was enqueued by ONE C call (~5 ms of host time for ~19 ms of GPU work); the collectives are enqueued now.  ``enqueue_lead`` records,
        # per bucket, whether its gradient-group event had NOT fired yet when its wait + collective were enqueued - i.e. whether the host was ahead of
        # the GPU and the all-reduce can start the moment the gradients exist (False = the bucket was already complete: its collective starts late by
        # however long ago that was).  tools/dp_enqueue_lead.py prints it for the bench workload.
        self.enqueue_lead = []
        with torch.cuda.stream(self.comm_stream):
            for a, b, last in self._buckets:
                self.enqueue_lead.append(not self._events[last].query())
                self.comm_stream.wait_event(self._events[last])
                if self.grad_dtype == "bf16":
                    allreduce_mean_bf16_(flat[a:b], self.group, force)  # cast -> AVG -> cast back, all ordered by the communication stream
                else:
                    works.append(allreduce_mean_(flat[a:b], self.group, force))
        with torch.cuda.stream(main):
            for w in works:
                if w is not None:
                    w.wait()  # compute stream waits for the collective, the host does not
        main.wait_stream(self.comm_stream)
