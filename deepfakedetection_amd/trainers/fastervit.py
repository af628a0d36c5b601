"""FasterViT trainer on the MI355X engine — counterpart of the reference's trainers/fastervit.py.

Same `main()` contract; phases: head-only warm-up epoch (names containing "head", :396-432), then everything
trainable (:434-435) with micro-batches of 32 and 4 accumulation steps, both hard-coded in the reference
(:437-453), cosine LR, early stop on EARLY_STOP_PATIENCE (:322, :526), accuracy-only evaluation,
`FasterVitModel.pth` / latest.ckpt / best.ckpt.  The reference hard-codes MODEL_NAME = "faster_vit_2_224" (:62);
this trainer honours the orchestrator's MODEL_NAME so that BASELINE's FasterViT-0 trains through the same path.
The loop body lives in trainers/_engine.py.
"""

from __future__ import annotations

from ._engine import TrainerSpec, evaluate, run, train_one_epoch  # noqa: F401

MODEL_NAME = "faster_vit_2_224"
DEFAULT_EPOCHS, DEFAULT_BATCH_SIZE, DEFAULT_IMG_SIZE, DEFAULT_NUM_WORKERS = 25, 64, 224, 8
HEAD_LR, HEAD_WD, FT_LR, FT_WD = 3e-4, 5e-2, 1e-4, 5e-2
DEFAULT_PATIENCE = 4
BEST_WEIGHTS_NAME, BEST_CKPT_NAME, LATEST_CKPT_NAME = "FasterVitModel.pth", "best.ckpt", "latest.ckpt"
FT_BATCH_SIZE, FT_ACCUM_STEPS = 32, 4

SPEC = TrainerSpec(
    model_name=MODEL_NAME, best_weights_name=BEST_WEIGHTS_NAME, default_epochs=DEFAULT_EPOCHS,
    default_batch_size=DEFAULT_BATCH_SIZE, warmup_keys=("head",), unfreeze_keys=None, ft_batch_size=FT_BATCH_SIZE,
    ft_accum_steps=FT_ACCUM_STEPS, zero_grad_first=False, early_stop=True, default_patience=DEFAULT_PATIENCE,
    default_img_size=DEFAULT_IMG_SIZE, default_num_workers=DEFAULT_NUM_WORKERS, head_lr=HEAD_LR, head_wd=HEAD_WD, ft_lr=FT_LR,
    ft_wd=FT_WD,
)


def main() -> None:
    run(SPEC)


if __name__ == "__main__":
    main()
