"""EfficientFormerV2 trainer on the MI355X engine — counterpart of the reference's trainers/efficientformer_v2.py.

Same `main()` contract (no arguments, configuration through the environment: orchestrator.py:286-307), same phases:
head-only warm-up epoch (names containing "classifier" or "head", :349-387 — that substring also selects
`talking_head1/2` and `head_dist`), then fine-tuning of the parameters whose names contain any of UNFREEZE_KEYS
(:66-74, :389-393: the backward pass stops at the earliest such parameter, stages.2.blocks.3), BATCH_SIZE batches,
no accumulation, no early stop, accuracy-only evaluation, `EfficientFormerV2_S1.pth` / latest.ckpt / best.ckpt.
The loop body lives in trainers/_engine.py (shared with the FasterViT trainer); `TrainerSpec` below is the list of
what this script does differently from the EfficientNet one.
"""

from __future__ import annotations

from ._engine import TrainerSpec, evaluate, run, train_one_epoch  # noqa: F401  (re-exported: the reference exposes them)

MODEL_NAME = "efficientformerv2_s1"
DEFAULT_EPOCHS, DEFAULT_BATCH_SIZE, DEFAULT_IMG_SIZE, DEFAULT_NUM_WORKERS = 5, 128, 224, 8
DEFAULT_LR, DEFAULT_WEIGHT_DECAY = 1e-4, 5e-2
BEST_WEIGHTS_NAME, BEST_CKPT_NAME, LATEST_CKPT_NAME = "EfficientFormerV2_S1.pth", "best.ckpt", "latest.ckpt"
UNFREEZE_KEYS = ("stages.3", "blocks.3", "layer4", "bneck", "features.6", "classifier", "head")

SPEC = TrainerSpec(
    model_name=MODEL_NAME, best_weights_name=BEST_WEIGHTS_NAME, default_epochs=DEFAULT_EPOCHS,
    default_batch_size=DEFAULT_BATCH_SIZE, warmup_keys=("classifier", "head"), unfreeze_keys=UNFREEZE_KEYS,
    ft_batch_size=None, zero_grad_first=True, early_stop=False, default_img_size=DEFAULT_IMG_SIZE,
    default_num_workers=DEFAULT_NUM_WORKERS, ft_lr=DEFAULT_LR, ft_wd=DEFAULT_WEIGHT_DECAY, pass_img_size=True,
)


def main() -> None:
    run(SPEC)


if __name__ == "__main__":
    main()
