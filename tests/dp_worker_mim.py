"""Helper process of tests/test_accum_dp_gpu.py (not a test module): one rank of a ``torch.distributed.run``-style launch of
``smb_vision_amd.run_mim``.  Builds the trainer exactly as ``main`` does, trains, and writes what the parent test asserts on."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from smb_vision_amd.run_mim import build_trainer  # noqa: E402


def run(argv, dump):
    trainer = build_trainer(argv)
    model = trainer.model
    initial = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    seen = []
    stock = trainer.training_step

    def spy(m, inputs, *a, **k):
        if not seen:
            seen.append(float(inputs["pixel_values"].double().sum()))
            seen.append(type(m).__name__)
        return stock(m, inputs, *a, **k)

    trainer.training_step = spy
    trainer.train()
    torch.cuda.synchronize()
    opt = getattr(trainer.optimizer, "optimizer", trainer.optimizer)
    meta = {"wrapped": trainer.model_wrapped is not model or seen[1] != type(model).__name__, "dp_world": trainer.dp_engine.world if trainer.dp_engine else 1,
            "optimizer": type(opt).__name__, "first_batch_sum": seen[0]}
    torch.save({"initial": initial, "final": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "meta": meta}, dump)


if __name__ == "__main__":
    run(sys.argv[2:], sys.argv[1])
