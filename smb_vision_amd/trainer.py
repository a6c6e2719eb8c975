"""``SmbTrainer`` — HF ``Trainer`` with the MI355X data-parallel and optimizer tail wired in.

What the reference gets from ``accelerate launch --config_file acc_configs/multi_gpu_config.yaml`` (``scripts/training/run_vjepa.sh:16``,
``acc_configs/multi_gpu_config.yaml:1-17``: ``distributed_type: MULTI_GPU`` -> accelerate wraps the model in
``DistributedDataParallel``, 25 MB NCCL buckets) is replaced here, inside the same ``Trainer`` skeleton the reference scripts use
(``src/run_mim.py:429-436``, ``src/run_vjepa.py:249-256``):

* the model is NOT wrapped: ``accelerator.prepare_model`` hands a smb_vision_amd model back as is, and ``smb_vision_amd.dp.DataParallelEngine``
  is attached to it instead (RCCL all-reduce of the flat gradient arena on a side HIP stream, gated per layer group by HIP events the backward
  C call records — ``dp.py``).  Rank 0's parameters are broadcast at start, as DDP does at wrap time;
* the optimizer is ``SmbvAdamW`` (global-norm clip + AdamW as two HIP launches, ``optim.py``) with HF's decay / no-decay grouping, unless the
  caller passed ``optimizers=`` or asked for a non-AdamW ``--optim``; ``--max_grad_norm`` is honoured INSIDE its ``step`` and the logged
  ``grad_norm`` is the same pre-clip global norm HF reports, handed over as a device scalar (no per-step host sync);
* everything else (sampler sharding, schedule, logging, checkpointing, ``save_model``) is the stock ``Trainer``.

Launch exactly as the contract says: one process per GPU (``python -m torch.distributed.run --nproc-per-node N -m smb_vision_amd.run_mim ...``);
``torch.distributed`` backend "nccl" is RCCL on ROCm.  With gradient accumulation every micro-batch's gradients are averaged when they are
produced (accelerate's ``sync_each_batch`` behaviour): mean of sums == sum of means, so the result equals DDP's.
"""
from __future__ import annotations

import os
import re

import torch
import torch.distributed as dist
from transformers import Trainer

from .dp import DataParallelEngine
from .optim import SmbvAdamW

_NATIVE_OPTIMS = ("adamw_torch", "adamw_torch_fused", "adamw_hf")
_NORM_COMPONENT = re.compile(r"(?:^|\.)(?:layer_?)?norm(?:_?\w*)?\.(?:weight|bias)$", re.IGNORECASE)  # ...norm1.weight, ...layernorm_before.bias, ...


class DevicePrefetcher:
    """Wraps the training DataLoader: batch i+1 is uploaded on a copy stream while step i computes (the reference's Trainer uploads every 335 MB
    fp32 volume on the compute stream, ~7 ms of idle matrix cores per step at PCIe Gen5 rates).  Tensors arrive on the device; the consumer's
    stream waits on the upload's event, the host never blocks.  Page-locked source tensors copy asynchronously, pageable ones synchronously
    (still off the compute stream).  Everything else about the loader (length, sampler, set_epoch, ...) is forwarded."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def _upload(self, batch):
        def move(x):
            if torch.is_tensor(x):
                return x.to(self.device, non_blocking=True)
            if isinstance(x, dict):
                return {k: move(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(move(v) for v in x)
            return x
        with torch.cuda.stream(self.stream):
            out = move(batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev, batch  # (the host batch stays referenced until its copy has been waited for)

    def __iter__(self):
        it = iter(self.loader)
        try:
            pending = self._upload(next(it))
        except StopIteration:
            return
        while pending is not None:
            out, ev, _keep = pending
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)

            def mark(x):  # the caching allocator must not hand this memory to the copy stream's next batch while the step still reads it
                if torch.is_tensor(x):
                    x.record_stream(cur)
                elif isinstance(x, dict):
                    for v in x.values():
                        mark(v)
                elif isinstance(x, (list, tuple)):
                    for v in x:
                        mark(v)
            mark(out)
            yield out
            pending = nxt


class VolumeFeedLoader:
    """Wraps a DataLoader whose batches carry RAW volumes (``feed.RAW_KEY``: tensors, arrays or ``volcache.VolumeRecord`` - what ``run_mim`` / ``run_vjepa``
    build under ``--volume_cache_dir``, the role of the reference's ``--json_path`` / ``--cache_dir``: ``src/run_mim.py:61, 140-143, 279-290``) and uploads
    them through ``feed.VolumeFeeder``: files are read straight into the feeder's page-locked ring by its host thread, travel over PCIe in their storage
    dtype on the copy stream and are scaled / padded / cropped / permuted on the device (``smbv_prepare_volume``) one batch ahead of the step.  Yields
    the loader's batch with ``out_key`` = fp32 ``[B, depth, 1, image_size, image_size]`` in place of the raw volumes and every other tensor on the device.
    Length, sampler, ``set_epoch`` ... are the wrapped loader's."""

    def __init__(self, loader, device, image_size, depth, out_key="pixel_values"):
        from .feed import VolumeFeeder
        self.loader, self.device, self.out_key = loader, torch.device(device), out_key
        # fresh outputs: HF fetches every micro-batch of an accumulation step before computing the first, so several batches are alive at once
        self.feeder = VolumeFeeder((), self.device, image_size, depth, record_window=True, fresh_outputs=True)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):
        return getattr(self.loader, name)

    def __iter__(self):
        from .feed import RAW_KEY

        def source():
            for batch in self.loader:
                batch = dict(batch)
                yield batch.pop(RAW_KEY), batch

        def move(x):
            if torch.is_tensor(x):
                return x.to(self.device, non_blocking=True)
            if isinstance(x, dict):
                return {k: move(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(move(v) for v in x)
            return x

        self.feeder.source = source()
        for pix, rest in self.feeder:
            out = move(rest)          # masks and index lists: a few hundred KB on the compute stream
            out[self.out_key] = pix
            yield out


def _is_native(model) -> bool:
    return hasattr(model, "flat_grads") and hasattr(model, "grad_group_ranges")


class SmbTrainer(Trainer):
    def __init__(self, *args, dp_bucket_bytes: int = 25 << 20, dp_grad_dtype: str = None, **kwargs):
        self._dp_bucket_bytes = dp_bucket_bytes
        # what the gradient all-reduce ships: "fp32" (default, DDP's plain buckets) or "bf16" (DDP's bf16_compress_hook: half the xGMI bytes); SMBV_DP_GRAD_DTYPE
        self._dp_grad_dtype = dp_grad_dtype or os.environ.get("SMBV_DP_GRAD_DTYPE", "fp32")
        self.dp_engine = None
        super().__init__(*args, **kwargs)

    # -- no DDP wrap ----------------------------------------------------------------------------------------------------------------------
    def create_accelerator_and_postprocess(self) -> None:
        super().create_accelerator_and_postprocess()
        acc = self.accelerator
        stock = acc.prepare_model

        def prepare_model(model, device_placement=None, evaluation_mode=False):
            if _is_native(model):  # gradient averaging belongs to the model's DataParallelEngine
                if model not in acc._models:
                    acc._models.append(model)
                return model
            return stock(model, device_placement=device_placement, evaluation_mode=evaluation_mode)

        acc.prepare_model = prepare_model

    def _prepare_for_training(self, *args, **kwargs):
        out = super()._prepare_for_training(*args, **kwargs)
        model = self.model
        force = os.environ.get("SMBV_DP_FORCE_COLLECTIVES", "0") == "1"  # keep the collective path alive at world size 1 (a one-GPU rehearsal of the RCCL path)
        if _is_native(model) and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force) and self.dp_engine is None:
            overlap = "nccl" in str(dist.get_backend()).lower()  # the side-stream overlap needs RCCL's stream-ordered collectives
            self.dp_engine = DataParallelEngine(model, bucket_bytes=self._dp_bucket_bytes, overlap=overlap, force_collectives=force,
                                                grad_dtype=self._dp_grad_dtype)
            self.dp_engine.broadcast_parameters(model)
        return out

    def get_train_dataloader(self):
        prefetch = getattr(self.args, "smbv_prefetch_to_device", True) and _is_native(self.model) and next(self.model.parameters()).is_cuda
        if not prefetch:
            return super().get_train_dataloader()
        # accelerate's prepared loader would itself move every batch to the device with a BLOCKING copy on the compute stream
        # (DataLoaderShard.send_to_device): leave the batches on the host and let the prefetcher upload them on its copy stream
        placement = self.accelerator.device_placement
        self.accelerator.device_placement = False
        try:
            loader = super().get_train_dataloader()
        finally:
            self.accelerator.device_placement = placement
        device = next(self.model.parameters()).device
        if getattr(self.train_dataset, "smbv_raw_volumes", False):   # items are raw volumes / cache records: upload through the VolumeFeeder
            size, depth, key = self._volume_geometry()
            return VolumeFeedLoader(loader, device, size, depth, out_key=key)
        return DevicePrefetcher(loader, device)

    def _volume_geometry(self):
        """(image_size, depth, name of the model's volume argument) from the model's config: VideoMAE names them image_size / num_frames
        (``src/run_mim.py:302-330``), V-JEPA crop_size / frames_per_clip (``src/run_vjepa.py:221-233``)."""
        c = self.model.config
        if hasattr(c, "crop_size"):
            return int(c.crop_size), int(c.frames_per_clip), "pixel_values_videos"
        return int(c.image_size), int(c.num_frames), "pixel_values"

    def _prepare_inputs(self, inputs):
        """A batch that still carries raw volumes (evaluation; the first, partial epoch of a resumed run, for which HF rebuilds a plain DataLoader behind
        the wrapper): the feeder's conversion inline on the compute stream - same kernel, same bits, no overlap."""
        from .feed import RAW_KEY, raw_batch_to_device
        if isinstance(inputs, dict) and RAW_KEY in inputs:
            inputs = dict(inputs)
            size, depth, key = self._volume_geometry()
            inputs[key] = raw_batch_to_device(inputs.pop(RAW_KEY), next(self.model.parameters()).device, size, depth)
        return super()._prepare_inputs(inputs)

    def _refuse_after_device_fault(self, what: str) -> None:
        """Nothing of weights a faulted step has touched is written: the library's sticky status word is set by a kernel that detects a fault it
        cannot return (``smbv_device_status``); it is read here after a device synchronisation, BEFORE anything goes to disk - an auto-resume must never pick up
        a checkpoint, and ``output_dir`` never a ``model.safetensors``, whose last update consumed incomplete gradients.  Under data parallelism the ranks agree on
        the verdict first (MAX over the status words): a rank that raised alone would leave the others inside HF's save barrier until the collective timed out."""
        if not (_is_native(self.model) and next(self.model.parameters()).is_cuda):
            return
        from . import _lib, ops
        dev = next(self.model.parameters()).device
        with torch.cuda.device(dev):          # the status word is per device: the model's, whatever the process's current device is
            torch.cuda.synchronize(dev)
            status = ops.device_status()
        mine = _lib.last_error() if status != 0 else ""
        faulted = status != 0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            on_gpu = "nccl" in str(dist.get_backend()).lower()
            flag = torch.tensor([1 if faulted else 0], dtype=torch.int32, device=dev if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            faulted = bool(int(flag.item()))
        if faulted:
            where = mine or "recorded on another rank"
            raise _lib.SmbvError(f"refusing to write {what}: a kernel of an earlier step recorded a fault on this device ({where})")

    def _save_checkpoint(self, *args, **kwargs):
        self._refuse_after_device_fault("a checkpoint")
        return super()._save_checkpoint(*args, **kwargs)

    def save_model(self, *args, **kwargs):
        """``train_and_evaluate`` (and the reference scripts, ``src/run_mim.py:441``) call this right after ``train()``: a fault recorded by the LAST steps is seen by
        no later library entry, so it is looked for here as well."""
        self._refuse_after_device_fault("the model")
        return super().save_model(*args, **kwargs)

    # -- decay grouping -------------------------------------------------------------------------------------------------------------------
    def get_decay_parameter_names(self, model):
        """HF excludes LayerNorm parameters from weight decay by MODULE TYPE (``nn.LayerNorm``) or by name (``bias``, ``layernorm``, ``.norm.``).  The
        native models keep their parameters in plain containers named as the reference's state dict names them (``norm1`` / ``norm2`` for the V-JEPA
        blocks, ``src/models/vjepa/modeling_vjepa.py:426-478``), which match neither rule - so the LayerNorm gains are taken out here by their
        container name.  Pinned by ``tests/golden/decay_names.json`` (HF's own split of the reference classes)."""
        names = Trainer.get_decay_parameter_names(self, model)
        if not _is_native(model):
            return names
        return [n for n in names if not _NORM_COMPONENT.search(n)]

    # -- fused optimizer tail -------------------------------------------------------------------------------------------------------------
    def create_optimizer(self, model=None):
        opt_model = self.model if model is None else model
        if self.optimizer is None and self.optimizer_cls_and_kwargs is None and str(getattr(self.args.optim, "value", self.args.optim)) in _NATIVE_OPTIMS \
                and _is_native(opt_model):
            decay = self.get_decay_parameter_names(opt_model)
            groups = [{"params": [p for n, p in opt_model.named_parameters() if n in decay and p.requires_grad], "weight_decay": self.args.weight_decay},
                      {"params": [p for n, p in opt_model.named_parameters() if n not in decay and p.requires_grad], "weight_decay": 0.0}]
            groups = [g for g in groups if g["params"]]
            mgn = self.args.max_grad_norm if self.args.max_grad_norm is not None and self.args.max_grad_norm > 0 else None
            self.optimizer = SmbvAdamW(groups, lr=self.args.learning_rate, betas=(self.args.adam_beta1, self.args.adam_beta2), eps=self.args.adam_epsilon,
                                       max_grad_norm=mgn)
            return self.optimizer
        return super().create_optimizer(model)

    # -- evaluation -----------------------------------------------------------------------------------------------------------------------
    def prediction_step(self, model, inputs, prediction_loss_only, ignore_keys=None):
        """``trainer.evaluate()`` (src/run_mim.py:452-455).  The reference's models take no ``labels``: the stock step then calls ``model(**inputs)``, keeps every
        output as a "prediction" (for one 512 x 512 x 320 volume: 109 MB of bf16 reconstruction copied to the host per sample) and reports NO loss - and its
        ``compute_metrics`` (``:435``: the loss as first prediction) is never reached, because HF only calls it when label ids exist.  For the native models the
        step returns the model's own loss and nothing else: ``eval_loss`` is then HF's mean over the evaluation batches."""
        if not _is_native(model):
            return super().prediction_step(model, inputs, prediction_loss_only, ignore_keys=ignore_keys)
        inputs = self._prepare_inputs(inputs)
        with torch.no_grad():
            outputs = model(**inputs)
        loss = getattr(outputs, "loss", None)
        if loss is None:
            return super().prediction_step(model, inputs, prediction_loss_only, ignore_keys=ignore_keys)
        return (loss.detach().float(), None, None)

    def _native_optimizer(self):
        opt = self.optimizer
        opt = getattr(opt, "optimizer", opt)  # accelerate's AcceleratedOptimizer
        return opt if isinstance(opt, SmbvAdamW) else None

    def _clip_grad_norm(self, model):
        opt = self._native_optimizer()
        if opt is None or not opt.max_grad_norm:
            return super()._clip_grad_norm(model)
        return opt.compute_grad_norm()  # device scalar; the clip itself happens inside opt.step() with this same value


def unwrap_optimizer(optimizer):
    return getattr(optimizer, "optimizer", optimizer)


def detect_last_checkpoint(training_args, logger=None):
    """``src/run_mim.py:263-276`` / ``src/run_vjepa.py:191-205``: an ``output_dir`` that already holds ``checkpoint-*`` directories resumes from the newest one
    unless ``--resume_from_checkpoint`` names another (or ``--overwrite_output_dir``, where the installed transformers still has that flag)."""
    import os
    from transformers.trainer_utils import get_last_checkpoint
    last = None
    out = training_args.output_dir
    if out and os.path.isdir(out) and training_args.do_train and not getattr(training_args, "overwrite_output_dir", False):
        last = get_last_checkpoint(out)
        if last is None and len(os.listdir(out)) > 0 and hasattr(training_args, "overwrite_output_dir"):
            raise ValueError(f"Output directory ({out}) already exists and is not empty. Use --overwrite_output_dir to overcome.")
        if last is not None and training_args.resume_from_checkpoint is None and logger is not None:
            logger.info(f"Checkpoint detected, resuming training at {last}. To avoid this behavior, change the `--output_dir`.")
    return last


def train_and_evaluate(trainer, training_args, last_checkpoint=None):
    """The tail both reference scripts share (``src/run_mim.py:439-455``, ``src/run_vjepa.py:258-275``): ``train(resume_from_checkpoint=...)`` with the
    explicit checkpoint first, the detected one second; ``save_model`` / metrics / state; ``evaluate()`` under ``--do_eval``."""
    metrics = {}
    if training_args.do_train:   # src/run_mim.py:440 / src/run_vjepa.py:259: `--do_eval` alone evaluates, it neither trains nor overwrites output_dir
        checkpoint = training_args.resume_from_checkpoint if training_args.resume_from_checkpoint is not None else last_checkpoint
        result = trainer.train(resume_from_checkpoint=checkpoint)
        trainer.save_model()
        trainer.log_metrics("train", result.metrics)
        trainer.save_metrics("train", result.metrics)
        trainer.save_state()
        metrics.update(result.metrics)
    if training_args.do_eval and trainer.eval_dataset is not None:
        ev = trainer.evaluate()
        trainer.log_metrics("eval", ev)
        trainer.save_metrics("eval", ev)
        metrics.update(ev)
    return metrics


def loss_only_metrics(eval_pred):
    """The reference's ``compute_metrics=lambda eval_pred: {"loss": eval_pred.predictions[0].item()}`` (``src/run_mim.py:435``, ``src/run_vjepa.py:255``): the
    model's loss is the first "prediction".  ``.item()`` only works for ONE evaluation batch; the mean over the batches is the
    same number there and defined for more."""
    import numpy as np
    p = eval_pred.predictions
    p = p[0] if isinstance(p, (tuple, list)) else p
    return {"loss": float(np.mean(np.asarray(p, dtype=np.float64)))}


def keep_loss_only(logits, labels):
    """``preprocess_logits_for_metrics``: of (loss, logits [B, N_mask, 4096], ...) only the loss travels to the host - the reconstruction of ONE 512 x 512 x 320
    volume is 109 MB of bf16 that ``compute_metrics`` never reads."""
    first = logits[0] if isinstance(logits, (tuple, list)) else logits
    return first.detach().float().reshape(-1)[:1] if first.dim() == 0 or first.numel() == 1 else first

