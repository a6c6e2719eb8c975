#!/usr/bin/env python3
"""Classification fine-tuning entry point on the MI355X path - the harness of reference ``src/run_classification.py`` with the model class
swapped (``:15`` / ``:495``).

Same skeleton: ``HfArgumentParser`` -> dataset -> ``VideoMAEConfig`` patched with ``num_labels`` / ``problem_type`` /
``additional_features_size`` (``:455-480``) -> model -> ``collate_fn`` producing ``{"pixel_values", "labels"[, "additional_features"]}``
(``:227-272``) -> HF ``Trainer``.  The dataset here is synthetic and in-memory (NIfTI/MONAI ingestion is out of scope); task types:
``classification`` (cross-entropy), ``regression`` (MSE), ``multilabel_classification`` (BCE-with-logits), as the reference's head computes
them (``src/models/videomae/modeling_videomae.py:991-1010``).
"""
from __future__ import annotations

import logging
import os
import sys
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch.utils.data import Dataset
from transformers import HfArgumentParser, Trainer, TrainingArguments, VideoMAEConfig, set_seed

from .modeling_videomae_encoder import SmbVideoMAEForVideoClassification

logger = logging.getLogger(__name__)
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
os.environ.setdefault("WANDB_MODE", "disabled")


@dataclass
class DataTrainingArguments:
    image_size: int = 224
    depth: int = 160
    patch_size: int = 16
    task_type: str = field(default="classification", metadata={"help": "classification | regression | multilabel_classification"})
    num_labels: int = 2
    num_additional_features: int = field(default=0, metadata={"help": "len(additional_feature_columns) of the reference"})
    num_train_volumes: int = 16


@dataclass
class ModelArguments:
    model_name_or_path: Optional[str] = field(default=None, metadata={"help": "LOCAL checkpoint directory (e.g. a pre-trained MIM encoder)"})
    hidden_size: Optional[int] = None
    num_hidden_layers: Optional[int] = None
    num_attention_heads: Optional[int] = None
    intermediate_size: Optional[int] = None


class SyntheticLabelledVolumes(Dataset):
    def __init__(self, n, args: DataTrainingArguments, seed=4321):
        self.n, self.a, self.seed = n, args, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed + i)
        a = self.a
        item = {"image": torch.rand(a.depth, 1, a.image_size, a.image_size, generator=g)}
        if a.task_type == "regression":
            item["label"] = torch.rand((), generator=g)
        elif a.task_type == "multilabel_classification":
            item["label"] = (torch.rand(a.num_labels, generator=g) > 0.5).float()
        else:
            item["label"] = torch.randint(0, a.num_labels, (), generator=g)
        if a.num_additional_features:
            item["extra"] = torch.rand(a.num_additional_features, generator=g)
        return item


def collate_fn(examples):
    """``src/run_classification.py:227-272``."""
    out = {"pixel_values": torch.stack([e["image"] for e in examples]), "labels": torch.stack([e["label"] for e in examples])}
    if "extra" in examples[0]:
        out["additional_features"] = torch.stack([e["extra"] for e in examples])
    return out


def main(argv=None):
    parser = HfArgumentParser((ModelArguments, DataTrainingArguments, TrainingArguments))
    margs, dargs, targs = parser.parse_args_into_dataclasses(args=argv)
    logging.basicConfig(level=logging.INFO, handlers=[logging.StreamHandler(sys.stdout)])
    set_seed(targs.seed)
    targs.remove_unused_columns = False
    config = VideoMAEConfig(num_channels=1, num_frames=dargs.depth, tubelet_size=dargs.patch_size, image_size=dargs.image_size, patch_size=dargs.patch_size,
                            use_mean_pooling=True)
    for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size"):
        if getattr(margs, k) is not None:
            setattr(config, k, getattr(margs, k))
    # run_classification.py:455-480
    config.num_labels = 1 if dargs.task_type == "regression" else dargs.num_labels
    config.problem_type = {"regression": "regression", "multilabel_classification": "multi_label_classification"}.get(dargs.task_type, "single_label_classification")
    if dargs.num_additional_features:
        config.update({"additional_features_size": dargs.num_additional_features})
    if margs.model_name_or_path:  # e.g. the encoder of a MIM checkpoint: decoder weights are ignored, the head is newly initialised
        model = SmbVideoMAEForVideoClassification.from_pretrained(margs.model_name_or_path, config=config, ignore_mismatched_sizes=True)
    else:
        model = SmbVideoMAEForVideoClassification(config)
    trainer = Trainer(model=model, args=targs, train_dataset=SyntheticLabelledVolumes(dargs.num_train_volumes, dargs), data_collator=collate_fn)
    result = trainer.train()
    trainer.save_model()
    trainer.log_metrics("train", result.metrics)
    return result.metrics


if __name__ == "__main__":
    main()
