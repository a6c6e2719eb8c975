#!/usr/bin/env python3
"""MIM pre-training entry point on the MI355X path — the harness of reference ``src/run_mim.py`` with the model class swapped.

Same skeleton as the reference script (``src/run_mim.py:222-471``): ``HfArgumentParser`` over (Model, Data, Training) argument
dataclasses → dataset → ``VideoMAEConfig`` patched with ``num_channels=1, num_frames=depth, tubelet_size=patch_size, image_size``
(``:302-330``) → model → ``collate_fn`` producing ``{"pixel_values", "bool_masked_pos"}`` (``:194-219``) → HF ``Trainer`` →
``train`` / ``save_model`` / metrics.  Differences, all on purpose:

* the model is ``SmbVideoMAEForPreTraining`` (HIP kernels behind the same interface) and the ``Trainer`` is ``SmbTrainer`` (``trainer.py``): under a
  one-process-per-GPU launch it averages gradients over RCCL itself (replacing ``acc_configs/multi_gpu_config.yaml``) and runs the fused optimizer tail;
* the dataset is either ``--volume_cache_dir`` (a directory of ``.smbvol`` files, ``volcache.py``: raw int16 volumes read into the feeder's page-locked ring,
  uploaded and scaled / padded / cropped / permuted on the device - the role of ``--json_path`` / ``--cache_dir``, ``src/run_mim.py:61, 140-143, 279-290``)
  or synthetic and in-memory (``SyntheticVolumeDataset``: U[0,1) volumes in the post-``ScaleIntensityRanged`` range, ``(D,1,H,W)`` layout of
  ``PermuteImage``, masks from the bit-exact ``MaskGenerator``): NIfTI reading / resampling is out of scope (nibabel / MONAI are absent
  from this image); any dataset returning the reference's two keys (``image``, ``mask``) plugs in as well;
* nothing touches the network (no telemetry, no hub names: pass local paths only).

    python -m smb_vision_amd.run_mim --output_dir /tmp/mim --image_size 64 --depth 32 --patch_size 16 --mask_patch_size 16 \
        --mask_ratio 0.6 --max_steps 10 --per_device_train_batch_size 2 --report_to none --bf16 true
"""
from __future__ import annotations

import logging
import os
import sys
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
from torch.utils.data import Dataset
from transformers import HfArgumentParser, TrainingArguments, VideoMAEConfig, set_seed

from .masking import MaskGenerator, split_mask_indices_host
from .modeling_videomae import SmbVideoMAEForPreTraining
from .trainer import SmbTrainer, detect_last_checkpoint, keep_loss_only, loss_only_metrics, train_and_evaluate

logger = logging.getLogger(__name__)
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
os.environ.setdefault("WANDB_MODE", "disabled")
# RCCL's kernels on high-priority streams (read when the process group is created, i.e. when TrainingArguments first asks for its device): the gradient
# all-reduce is dispatched ahead of the backward's queued 0.5-2 ms attention blocks instead of behind them (dp.py)
os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")


@dataclass
class DataTrainingArguments:
    """Subset of the reference's DataTrainingArguments that shapes the hot path (``src/run_mim.py:45-110``)."""

    image_size: int = field(default=224, metadata={"help": "The size (resolution) of each image."})
    depth: int = field(default=160, metadata={"help": "The depth of each volume."})
    patch_size: int = field(default=16, metadata={"help": "The size (resolution) of each patch."})
    mask_patch_size: int = field(default=32, metadata={"help": "The size of the square patches to use for masking."})
    mask_ratio: float = field(default=0.5, metadata={"help": "Percentage of patches to mask."})
    num_train_volumes: int = field(default=16, metadata={"help": "Synthetic volumes per epoch."})
    num_eval_volumes: int = field(default=2, metadata={"help": "Synthetic validation volumes (used with --do_eval / --eval_strategy)."})
    data_seed_offset: int = field(default=1234, metadata={"help": "Seed of the synthetic volumes."})
    volume_cache_dir: Optional[str] = field(default=None, metadata={"help": "Directory of .smbvol files (smb_vision_amd.volcache) to train on instead of synthetic "
                                            "volumes: the role of the reference's --json_path / --cache_dir (src/run_mim.py:61, 140-143, 279-290)."})
    eval_volume_cache_dir: Optional[str] = field(default=None, metadata={"help": "Directory of .smbvol files for evaluation (else synthetic validation volumes)."})
    fresh_masks: bool = field(default=False, metadata={"help": "With --volume_cache_dir: draw a new mask per visit from the global numpy stream instead of one "
                                                       "fixed mask per sample (the reference's PersistentDataset caches the mask with the image)."})


@dataclass
class ModelArguments:
    """``src/run_mim.py:113-192`` (local paths only)."""

    model_name_or_path: Optional[str] = field(default=None, metadata={"help": "LOCAL checkpoint directory to start from."})
    config_name_or_path: Optional[str] = field(default=None, metadata={"help": "LOCAL config.json directory (else VideoMAEConfig())."})
    hidden_size: Optional[int] = None
    num_hidden_layers: Optional[int] = None
    num_attention_heads: Optional[int] = None
    intermediate_size: Optional[int] = None
    decoder_hidden_size: Optional[int] = None
    decoder_num_hidden_layers: Optional[int] = None
    decoder_num_attention_heads: Optional[int] = None
    decoder_intermediate_size: Optional[int] = None


class SyntheticVolumeDataset(Dataset):
    """Items shaped like the reference pipeline's output (``src/dataloader/mim.py:142-185``): ``image`` f32 (D,1,H,W) in [0,1],
    ``mask`` bool (N,).  The mask of item i is drawn from the legacy numpy stream seeded with (seed + i) - a fixed mask per sample, which is what the
    reference's ``PersistentDataset`` ends up with (``GenerateMask`` is a deterministic transform, so the mask is cached with the image: SURVEY.md 8f-3) and
    what makes a resumed run see the batches the uninterrupted run saw (HF skips batches by index, without producing the items).  The global numpy
    stream is left as it was found."""

    def __init__(self, n, image_size, depth, mask_generator, seed):
        self.n, self.image_size, self.depth, self.gen, self.seed = n, image_size, depth, mask_generator, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        state = np.random.get_state()
        np.random.seed((self.seed + i) % (2 ** 32))
        try:
            mask = self.gen()
        finally:
            np.random.set_state(state)
        return {"image": torch.rand(self.depth, 1, self.image_size, self.image_size, generator=g), "mask": mask}


def unwrap_examples(examples):
    """``src/run_mim.py:195-202`` / ``src/run_vjepa.py:145-152``: MONAI datasets hand items over wrapped in singleton lists (``[item]``, ``[[item]]``);
    peel them until the dictionary appears."""
    out = []
    for ex in examples:
        while isinstance(ex, (list, tuple)) and len(ex) == 1:
            ex = ex[0]
        out.append(ex)
    return out


def collate_fn(examples):
    """``src/run_mim.py:194-219``.  Items that carry a raw cached volume (``"volume"``: ``volcache.CachedVolumeDataset``) instead of an ``"image"`` travel on
    as ``feed.RAW_KEY``; the trainer's loader turns them into ``pixel_values`` on the device (``trainer.VolumeFeedLoader``)."""
    examples = unwrap_examples(examples)
    mask = torch.stack([e["mask"] for e in examples])
    vis, msk = split_mask_indices_host(mask)
    if "image" not in examples[0] and "volume" in examples[0]:
        from .feed import RAW_KEY
        return {RAW_KEY: [e["volume"] for e in examples], "bool_masked_pos": mask, "visible_index": vis, "masked_index": msk}
    # (a batch of one is a view, not a 335 MB copy: a page-locked volume stays page-locked for the upload)
    pixel_values = examples[0]["image"].unsqueeze(0) if len(examples) == 1 else torch.stack([e["image"] for e in examples])
    # extra keys for the HIP path: the two ascending token lists the reference derives with boolean indexing (spec :134-137, :811-812),
    # built here on the host from the mask that was just generated (the model otherwise sorts on the device every step)
    return {"pixel_values": pixel_values, "bool_masked_pos": mask, "visible_index": vis, "masked_index": msk}


def build_trainer(argv=None):
    """Everything of ``src/run_mim.py:222-436`` up to and including the ``Trainer`` construction."""
    parser = HfArgumentParser((ModelArguments, DataTrainingArguments, TrainingArguments))
    model_args, data_args, training_args = parser.parse_args_into_dataclasses(args=argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(name)s - %(message)s", level=logging.INFO, handlers=[logging.StreamHandler(sys.stdout)])
    set_seed(training_args.seed)
    training_args.remove_unused_columns = False
    # HF's default logging_nan_inf_filter reads the step's loss on the HOST every step (torch.isnan(loss) in the training loop): a device
    # synchronisation per step, which leaves the GPU idle while the host enqueues the next step's ~1,200 launches (~10 ms at ViT-base,
    # 512x512x320).  It only changes how a NaN step enters the LOGGED loss average; switched off here so that steps are enqueued back to back.
    training_args.logging_nan_inf_filter = False

    gen = MaskGenerator(input_size=data_args.image_size, depth=data_args.depth, mask_patch_size=data_args.mask_patch_size,
                        model_patch_size=data_args.patch_size, mask_ratio=data_args.mask_ratio)
    np.random.seed(training_args.seed)
    if not training_args.do_train and not training_args.do_eval:
        training_args.do_train = True   # a bare invocation trains, as this harness always did; `--do_eval true` alone only evaluates (src/run_mim.py:439, 452)
    if data_args.volume_cache_dir:      # real volumes from disk: raw int16 over PCIe, scaled / padded / cropped / permuted on the device (feed.py)
        from .volcache import CachedVolumeDataset
        train = CachedVolumeDataset(data_args.volume_cache_dir, mask_generator=gen, mask_seed=None if data_args.fresh_masks else data_args.data_seed_offset)
        train.smbv_raw_volumes = True
    else:
        train = SyntheticVolumeDataset(data_args.num_train_volumes, data_args.image_size, data_args.depth, gen, data_args.data_seed_offset)
    evaluating = training_args.do_eval or str(getattr(training_args.eval_strategy, "value", training_args.eval_strategy)) != "no"
    val = None
    if evaluating and data_args.eval_volume_cache_dir:
        from .volcache import CachedVolumeDataset
        val = CachedVolumeDataset(data_args.eval_volume_cache_dir, mask_generator=gen, mask_seed=data_args.data_seed_offset + 1_000_003)
    elif evaluating:
        val = SyntheticVolumeDataset(data_args.num_eval_volumes, data_args.image_size, data_args.depth, gen, data_args.data_seed_offset + 1_000_003)

    config = VideoMAEConfig.from_pretrained(model_args.config_name_or_path) if model_args.config_name_or_path else VideoMAEConfig()
    config.update({"num_channels": 1, "num_frames": data_args.depth, "tubelet_size": data_args.patch_size, "image_size": data_args.image_size,
                   "patch_size": data_args.patch_size})
    for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "decoder_hidden_size",
              "decoder_num_hidden_layers", "decoder_num_attention_heads", "decoder_intermediate_size"):
        if getattr(model_args, k) is not None:
            setattr(config, k, getattr(model_args, k))
    if model_args.model_name_or_path:
        model = SmbVideoMAEForPreTraining.from_pretrained(model_args.model_name_or_path, config=config)
    else:
        logger.info("Training new model from scratch")
        model = SmbVideoMAEForPreTraining(config)

    # SmbTrainer == HF Trainer + RCCL gradient averaging (instead of accelerate's DDP wrap, acc_configs/multi_gpu_config.yaml) + the fused clip/AdamW tail
    # eval_dataset / compute_metrics as src/run_mim.py:429-436 (HF reaches compute_metrics only when label ids exist - never for these models, in the
    # reference either; eval_loss comes from SmbTrainer.prediction_step)
    return SmbTrainer(model=model, args=training_args, train_dataset=train, eval_dataset=val, data_collator=collate_fn, compute_metrics=loss_only_metrics,
                      preprocess_logits_for_metrics=keep_loss_only)


def main(argv=None):
    """``src/run_mim.py:263-276, 439-455``: auto-resume from the newest ``checkpoint-*`` of ``output_dir`` (or ``--resume_from_checkpoint``), train, save,
    evaluate under ``--do_eval``."""
    trainer = build_trainer(argv)
    last = detect_last_checkpoint(trainer.args, logger)
    return train_and_evaluate(trainer, trainer.args, last)


if __name__ == "__main__":
    main()
