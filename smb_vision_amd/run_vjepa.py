#!/usr/bin/env python3
"""V-JEPA pre-training entry point on the MI355X path — the harness of reference ``src/run_vjepa.py`` with the model class swapped.

Mirrors ``src/run_vjepa.py:101-160, 163-290``: ``VJEPATrainer`` (frozen EMA target encoder created from the online model, L1 loss
between the predictor output and the target encoder's target tokens, EMA update inside ``compute_loss``), the collate function that
broadcasts ONE example's masks to the whole batch, and the ``HfArgumentParser`` / ``Trainer`` skeleton.  The dataset is synthetic and
in-memory (NIfTI/MONAI ingestion is out of scope); the model, loss and EMA run on the HIP path (``modeling_vjepa.py``)."""
from __future__ import annotations

import logging
import os
import functools
import random
import sys
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch.utils.data import Dataset
from transformers import HfArgumentParser, TrainingArguments, VJEPA2Config, set_seed

from .masking import VJEPAMaskGenerator
from .modeling_vjepa import MomentumEncoder, SmbVJEPA2Model, l1_loss, make_target_encoder, vjepa_loss
from .trainer import SmbTrainer, detect_last_checkpoint, keep_loss_only, loss_only_metrics, train_and_evaluate

logger = logging.getLogger(__name__)
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
os.environ.setdefault("WANDB_MODE", "disabled")
# RCCL's kernels on high-priority streams (read when the process group is created, i.e. when TrainingArguments first asks for its device): the gradient
# all-reduce is dispatched ahead of the backward's queued 0.5-2 ms attention blocks instead of behind them (dp.py)
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
random.seed(42)  # run_vjepa.py:30


class SmbVJEPATrainer(SmbTrainer):
    """``VJEPATrainer`` (run_vjepa.py:101-141) on top of ``SmbTrainer`` (RCCL gradient averaging of the ONLINE model only — the target encoder is a
    function of the online parameters, identical on every rank — and the fused optimizer tail)."""

    def __init__(self, *args, momentum=0.99925, **kwargs):
        super().__init__(*args, **kwargs)
        self.target_encoder = None
        self.momentum = momentum
        self.momentum_encoder = None
        # The reference assigns ``self.compute_loss_func = nn.L1Loss()`` (run_vjepa.py:108); with this transformers version that attribute switches
        # OFF the division of the loss by gradient_accumulation_steps in training_step.  Reproduced: any non-None
        # value has that effect, and compute_loss below never calls it.
        self.compute_loss_func = l1_loss

    def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None):
        if self.target_encoder is None:  # created once the online model sits on its device
            self.target_encoder = make_target_encoder(getattr(model, "module", model))
            self.momentum_encoder = MomentumEncoder(self.target_encoder, momentum=self.momentum)
        loss, outputs = vjepa_loss(model, self.target_encoder, self.momentum_encoder, inputs)
        return (loss, outputs) if return_outputs else loss

    def prediction_step(self, model, inputs, prediction_loss_only, ignore_keys=None):
        """Evaluation (``trainer.evaluate()``, run_vjepa.py:270-275): the stock step would call ``model(**inputs)``, whose output has no loss at all - the L1
        loss lives in ``compute_loss``.  Here: the same loss against the current target encoder, no gradient, NO momentum update."""
        inputs = self._prepare_inputs(inputs)
        if self.target_encoder is None:
            self.target_encoder = make_target_encoder(getattr(model, "module", model))
            self.momentum_encoder = MomentumEncoder(self.target_encoder, momentum=self.momentum)
        with torch.no_grad():
            loss, _ = vjepa_loss(model, self.target_encoder, None, inputs)
        loss = loss.detach()
        return (loss, None, None) if prediction_loss_only else (loss, loss.reshape(1), None)


def collate_fn(examples, num_patches=0):
    """run_vjepa.py:144-160: stack the volumes; ONE random example's masks for every row; masks wrapped in 1-element lists.
    ``num_patches`` (the model's token count; the harness binds it with functools.partial): also ship the step's index arrays, see below."""
    from .run_mim import unwrap_examples
    examples = unwrap_examples(examples)   # run_vjepa.py:145-152: singleton-list wrappers of MONAI datasets
    ex = random.choice(examples)
    context_mask = torch.stack([ex["context_mask"] for _ in examples])
    target_mask = torch.stack([ex["target_mask"] for _ in examples])
    if "image" not in examples[0] and "volume" in examples[0]:   # raw cached volumes (--volume_cache_dir): converted on the device by the trainer's loader
        from .feed import RAW_KEY
        batch = {RAW_KEY: [e["volume"] for e in examples]}
    else:
        batch = {"pixel_values_videos": torch.stack([e["image"] for e in examples])}
    batch.update({"context_mask": [context_mask], "target_mask": [target_mask]})
    n_tokens = num_patches
    if n_tokens and context_mask.shape[1] < n_tokens and target_mask.shape[1] < n_tokens:
        # the index arithmetic of the step (sorted predictor slots, the encoder's [consumed | other] token order) done here, on the host, by the
        # DataLoader's workers: vjepa_loss hands each array set to the forward it was built for (modeling_vjepa.index_plumbing)
        from .modeling_vjepa import index_plumbing
        batch["smbv_plumbing_context"] = index_plumbing([context_mask.long()], [target_mask.long()], n_tokens, "context")
        batch["smbv_plumbing_target"] = index_plumbing([context_mask.long()], [target_mask.long()], n_tokens, "target")
    return batch



class SyntheticVJEPADataset(Dataset):
    def __init__(self, n, crop, depth, gen, seed):
        self.n, self.crop, self.depth, self.gen, self.seed = n, crop, depth, gen, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        item = {"image": torch.rand(self.depth, 1, self.crop, self.crop, generator=g)}
        # a fixed mask per sample (the generator draws from torch's GLOBAL stream, transforms.py:168-170): what the reference's PersistentDataset caches
        # with the image, and what lets a resumed run see the uninterrupted run's batches; the global stream is left as it was found
        # (default_generator.manual_seed, not torch.manual_seed: the latter also reseeds every CUDA generator, which set_rng_state below would not put back)
        state = torch.random.get_rng_state()
        torch.default_generator.manual_seed(self.seed + i)
        try:
            return self.gen(item)
        finally:
            torch.random.set_rng_state(state)


class _SeededMasks:
    """``item_transform`` for ``volcache.CachedVolumeDataset``: the V-JEPA masks of a cached volume, one fixed pair per FILE (the generator draws from torch's
    GLOBAL stream, transforms.py:168-170; seeded by the file's name so that the pair does not depend on the visiting order; the stream is left as found)."""

    def __init__(self, gen, seed):
        self.gen, self.seed = gen, seed

    def __call__(self, item):
        import zlib
        state = torch.random.get_rng_state()
        torch.default_generator.manual_seed(self.seed + zlib.crc32(os.path.basename(item["volume"].path).encode()))   # CPU stream only, as above
        try:
            return self.gen(item)
        finally:
            torch.random.set_rng_state(state)


@dataclass
class DataTrainingArguments:
    image_size: int = field(default=224, metadata={"help": "crop size (H == W)"})
    depth: int = field(default=160, metadata={"help": "frames per clip"})
    patch_size: int = field(default=16)
    num_mask_blocks: int = field(default=3)
    num_train_volumes: int = field(default=16)
    num_eval_volumes: int = field(default=2)
    volume_cache_dir: Optional[str] = field(default=None, metadata={"help": "Directory of .smbvol files (smb_vision_amd.volcache) to train on instead of synthetic "
                                            "volumes: the role of the reference's --data_path / --cache_dir (src/run_vjepa.py:44, 69, 207-218)."})
    eval_volume_cache_dir: Optional[str] = field(default=None)


@dataclass
class ModelArguments:
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    pred_hidden_size: int = 384
    pred_num_hidden_layers: int = 12
    pred_num_attention_heads: int = 12


def main(argv=None):
    parser = HfArgumentParser((ModelArguments, DataTrainingArguments, TrainingArguments))
    margs, dargs, targs = parser.parse_args_into_dataclasses(args=argv)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(sys.stdout)])
    set_seed(targs.seed)
    targs.remove_unused_columns = False
    targs.logging_nan_inf_filter = False  # no per-step host read of the loss (see run_mim.py)
    # run_vjepa.py:221-233: in_chans=1, frames_per_clip=depth, tubelet_size=patch_size, crop_size=image_size
    config = VJEPA2Config(patch_size=dargs.patch_size, crop_size=dargs.image_size, frames_per_clip=dargs.depth, tubelet_size=dargs.patch_size, in_chans=1,
                          hidden_size=margs.hidden_size, num_hidden_layers=margs.num_hidden_layers, num_attention_heads=margs.num_attention_heads,
                          pred_hidden_size=margs.pred_hidden_size, pred_num_hidden_layers=margs.pred_num_hidden_layers,
                          pred_num_attention_heads=margs.pred_num_attention_heads)
    model = SmbVJEPA2Model(config)  # run_vjepa.py:245-246: always from scratch
    # grid in the model's token order (D', H', W'): see SURVEY.md Appendix A.8 on the reference generator's axis naming
    gen = VJEPAMaskGenerator(input_size=(dargs.depth, dargs.image_size, dargs.image_size), patch_size=(dargs.patch_size,) * 3,
                             pred_mask_scale=(0.2, 0.8), aspect_ratio=(0.3, 3.0), num_blocks=dargs.num_mask_blocks)
    if not targs.do_train and not targs.do_eval:
        targs.do_train = True   # a bare invocation trains; `--do_eval true` alone only evaluates (src/run_vjepa.py:259, 272)
    if dargs.volume_cache_dir:
        from .volcache import CachedVolumeDataset
        train = CachedVolumeDataset(dargs.volume_cache_dir, item_transform=_SeededMasks(gen, 1234))
        train.smbv_raw_volumes = True
    else:
        train = SyntheticVJEPADataset(dargs.num_train_volumes, dargs.image_size, dargs.depth, gen, 1234)
    evaluating = targs.do_eval or str(getattr(targs.eval_strategy, "value", targs.eval_strategy)) != "no"
    val = None
    if evaluating and dargs.eval_volume_cache_dir:
        from .volcache import CachedVolumeDataset
        val = CachedVolumeDataset(dargs.eval_volume_cache_dir, item_transform=_SeededMasks(gen, 1_001_237))
    elif evaluating:
        val = SyntheticVJEPADataset(dargs.num_eval_volumes, dargs.image_size, dargs.depth, gen, 1_001_237)
    # eval_dataset / compute_metrics as src/run_vjepa.py:249-256; auto-resume and the train / save / evaluate tail as :191-205, 258-275 (the EMA target
    # encoder lives on the trainer and is NOT part of a checkpoint - the reference's behaviour: a resumed run re-copies it from the restored online weights)
    trainer = SmbVJEPATrainer(model=model, args=targs, train_dataset=train, eval_dataset=val, data_collator=functools.partial(collate_fn, num_patches=model.num_patches),
                              compute_metrics=loss_only_metrics, preprocess_logits_for_metrics=keep_loss_only)
    last = detect_last_checkpoint(targs, logger)
    return train_and_evaluate(trainer, targs, last)


if __name__ == "__main__":
    main()
