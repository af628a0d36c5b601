"""Throughput of the drop-in trainer loop (`trainers.efficientnet.train_one_epoch`), eager dispatch vs hipGraph replay.

    python scripts/bench_trainer.py [--model efficientnet_b0] [--steps 60]

Runs the real loop body on synthetic pinned batches (so the loader's H2D copy is in, PIL decode is not) at the
reference's fine-tune configuration (micro-batch 32 x 4 accumulation steps, trainers/efficientnet.py:84-86) and at
batch 256 x 1, once with GRAPH_STEP off and once on, and prints one JSON line per case.  Numbers quoted in DESIGN.md.
"""

from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402
from rich.progress import Progress  # noqa: E402


class FakeLoader:
    """The attributes train_one_epoch touches on a DataLoader, over pre-pinned synthetic batches."""

    def __init__(self, batch: int, size: int, steps: int, classes: int) -> None:
        g = torch.Generator().manual_seed(1)
        self.batches = [(torch.randn(batch, 3, size, size, generator=g).pin_memory(), torch.randint(0, classes, (batch,), generator=g))
                        for _ in range(4)]
        self.batch_size, self.steps = batch, steps
        self.dataset = range(batch * steps)
        self.sampler = None

    def __len__(self) -> int:
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % 4]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="efficientnet_b0")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    from deepfakedetection_amd.graph_step import GraphedTrainStep
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss
    from deepfakedetection_amd.orchestration.model_registry import get_model_spec
    from deepfakedetection_amd.trainers.efficientnet import train_one_epoch

    for batch, accum in ((32, 4), (256, 1)):
        for graph in (False, True):
            torch.manual_seed(0)
            model = get_model_spec(args.model).builder(args.model, 2).cuda()
            opt = HipAdamW(model.parameters(), lr=1e-4, weight_decay=5e-2)
            crit = HipCrossEntropyLoss(0.1)
            scaler = torch.amp.GradScaler(enabled=False)
            stepper = GraphedTrainStep(model, crit, opt, accum_steps=accum) if graph else None
            steps = args.steps * accum
            with Progress(disable=True) as progress:
                warm = FakeLoader(batch, args.size, 3 * accum, 2)
                train_one_epoch(model, warm, opt, scaler, crit, "cuda", use_cuda_amp=True, progress=progress,
                                task=progress.add_task("w", total=len(warm)), accum_steps=accum, stepper=stepper)
                dl = FakeLoader(batch, args.size, steps, 2)
                stats: dict = {}
                loss = train_one_epoch(model, dl, opt, scaler, crit, "cuda", use_cuda_amp=True, progress=progress,
                                       task=progress.add_task("t", total=len(dl)), accum_steps=accum, stepper=stepper, stats=stats)
            print(json.dumps({"model": args.model, "micro_batch": batch, "accum_steps": accum, "requested": "hipgraph" if graph else "eager",
                              "launch": stats["launch"], "images_per_sec": round(stats["images_per_sec"], 1),
                              "ms_per_micro_batch": round(1e3 * stats["seconds"] / steps, 3), "mean_loss": round(loss, 4)}), flush=True)
            del model, opt, stepper
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
