"""The drop-in surface on the real device: `orchestrate(train yaml)` then
`orchestrate(inference yaml)` with the HIP-backed EfficientNet-B0 / -B3 on cuda, tiny
generated ImageFolder.  Everything between the YAML and the kernels is the product path."""

from __future__ import annotations

import json
from pathlib import Path

import pytest
import torch
import yaml

from tests.test_plumbing_cpu import _make_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_name,weights_file,gpu_tail", [("efficientnet_b0", "EfficientNetModel.pth", False),
                                                             ("efficientnet_b3", "EfficientNetModel.pth", False),
                                                             ("efficientnet_b0", "EfficientNetModel.pth", True),
                                                             ("efficientnet_b0", "EfficientNetModel.pth", "resize"),
                                                             ("efficientnet_b0", "EfficientNetModel.pth", "resize_default_toggles"),
                                                             ("efficientformerv2_s1", "EfficientFormerV2_S1.pth", False),
                                                             ("faster_vit_0_224", "FasterVitModel.pth", False)])
def test_orchestrated_training_and_inference_on_gpu(tmp_path, monkeypatch, model_name, weights_file, gpu_tail):
    from deepfakedetection_amd.orchestration.orchestrator import orchestrate

    monkeypatch.chdir(tmp_path)
    former = model_name.startswith("efficientformer")
    fvit = model_name.startswith("faster_vit")
    img = 224 if fvit else (128 if former else 64)   # EfficientFormerV2 at 128 px: 16-token attention; FasterViT: 7x7 windows need 224
    if gpu_tail == "resize_default_toggles":
        img = 224                                    # the reference's default pipeline: rotation + colour jitter ON above 64 pixels
    aug_calls = []
    if gpu_tail:
        from deepfakedetection_amd import kernels as KK

        real_augment = KK.augment_u8
        monkeypatch.setattr(KK, "augment_u8", lambda *a, **k: (aug_calls.append(1), real_augment(*a, **k))[1])
    _make_dataset(tmp_path / "data", classes=("fake", "real"), per_class=8, size=img + 8)
    base = {
        "seed": 1, "device": "cuda",
        "data": {"root": str(tmp_path / "data"), "train_split": "train", "val_split": "val", "test_split": "test",
                 "num_classes": 2, "img_size": img},
    }
    out_dir = str(tmp_path / "runs" / model_name)
    train_cfg = {**base, "models": {model_name: {"output_dir": out_dir, "training": {
        "epochs": 1, "batch_size": 8, "ft_batch_size": 8, "accum_steps": 2, "num_workers": 0, "resume": "auto", "pretrained": False,
        "gpu_input_tail": bool(gpu_tail), "gpu_resize": str(gpu_tail).startswith("resize")}}}}
    # gpu_tail True: loaders ship uint8, flip / normalise / erasing run in dfd_image_prep; "resize": the workers only decode and
    # plan, Resize / RandomResizedCrop / CenterCrop run in dfd_resize_crop_u8 (rotation and jitter off so that the TRAINING
    # pipeline qualifies too; the validation pipeline always does)
    if gpu_tail == "resize":
        train_cfg["models"][model_name]["transforms"] = {"train": {"train_color_jitter": False, "train_random_rotation": False}}
    path = tmp_path / "train.yaml"
    path.write_text(yaml.safe_dump(train_cfg))
    orchestrate(path, mode="training")
    if gpu_tail == "resize_default_toggles":
        # with the SHIPPED toggles (rotation and jitter on) the training batches took the device path —
        # dfd_resize_crop_u8 -> dfd_augment_u8 -> dfd_image_prep; warm-up + fine-tune epochs of 16 / 8 batches
        assert len(aug_calls) >= 4, "the default training pipeline did not run RandomRotation / ColorJitter on the device"
    elif gpu_tail:
        assert not aug_calls
    run = sorted(Path(out_dir).iterdir())[0]
    ckpt = torch.load(run / "checkpoints" / "latest.ckpt", map_location="cpu")
    assert ckpt["epoch"] == 1 and set(ckpt["optimizer"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    head_key = "head_dist.weight" if former else ("head.weight" if fvit else ("_fc.weight" if model_name == "efficientnet_b3" else "classifier.weight"))
    assert head_key in ckpt["model"]
    log = (run / "logs" / "train.log").read_text()
    assert "Warmup (head only)" in log and "val_acc=" in log
    if former:
        # fine-tuning trains the UNFREEZE_KEYS subset only (trainers/efficientformer_v2.py:389-393): 182 tensors
        assert len(ckpt["optimizer"]["state"]) == 182
        rows = [json.loads(line) for line in (run / "logs" / "throughput.jsonl").read_text().splitlines()]
        assert [r["phase"] for r in rows] == ["warmup", "fine-tune"] and all(r["images_per_sec"] > 0 for r in rows)

    infer_cfg = {**base, "models": {model_name: {"output_dir": out_dir, "inference": {
        # best weights exist only if the fine-tune epoch beat the warm-up (reference behaviour);
        # latest.ckpt always exists and load_model unwraps its "model" entry (orchestrator.py:370-374)
        "weights": str(run / "checkpoints" / "latest.ckpt"), "split": "test", "batch_size": 16, "num_workers": 0, "img_size": img}}}}
    path2 = tmp_path / "infer.yaml"
    path2.write_text(yaml.safe_dump(infer_cfg))
    orchestrate(path2, mode="inference")
    run2 = sorted(Path(out_dir).iterdir())[-1]
    row = json.loads((run2 / "logs" / "metrics.jsonl").read_text().splitlines()[0])
    assert row["model"] == model_name and 0.0 <= row["accuracy"] <= 1.0 and "threshold" in row
    # device-side resize for inference (`inference.gpu_resize: true`): bit-exact input pipeline -> the very same metrics row
    infer_cfg["models"][model_name]["inference"]["gpu_resize"] = True
    path2.write_text(yaml.safe_dump(infer_cfg))
    orchestrate(path2, mode="inference")
    run2b = sorted(Path(out_dir).iterdir())[-1]
    row_b = json.loads((run2b / "logs" / "metrics.jsonl").read_text().splitlines()[-1])
    assert row_b["accuracy"] == row["accuracy"] and row_b["confusion_matrix"] == row["confusion_matrix"]
    assert row_b.get("roc_auc") == row.get("roc_auc") and row_b.get("threshold") == row.get("threshold")
    infer_cfg["models"][model_name]["inference"]["gpu_resize"] = False
    # opt-in bf16 inference (`inference.amp: bf16`, an extra key of this engine): same plumbing
    infer_cfg["models"][model_name]["inference"]["amp"] = "bf16"
    path2.write_text(yaml.safe_dump(infer_cfg))
    orchestrate(path2, mode="inference")
    run3 = sorted(Path(out_dir).iterdir())[-1]
    row3 = json.loads((run3 / "logs" / "metrics.jsonl").read_text().splitlines()[-1])
    assert row3["model"] == model_name and 0.0 <= row3["accuracy"] <= 1.0
    assert sum(map(sum, row["confusion_matrix"])) == 16


def _build(family: str):
    if family == "efficientnet":
        from deepfakedetection_amd.efficientnet import HipEfficientNet

        return HipEfficientNet("b0", "timm", 2), 64
    if family == "efficientformer":
        from deepfakedetection_amd.efficientformer_v2 import build_efficientformer_v2

        return build_efficientformer_v2("efficientformerv2_s0", 2, 64), 64
    from deepfakedetection_amd.fastervit import build_fastervit

    return build_fastervit("faster_vit_0_224", 2), 224


@pytest.mark.parametrize("family,accum", [("efficientnet", 1), ("efficientnet", 3), ("efficientformer", 2), ("fastervit", 2)])
def test_graphed_step_is_bitwise_the_eager_step(family, accum):
    """graph_step.GraphedTrainStep replays exactly the kernels the eager loop body launches (same order, same
    fixed-order reductions, Philox masks from the same device-resident state): after several optimizer cycles the
    parameters, BatchNorm statistics and counters are bit-identical, also across an epoch boundary where the trainer
    drops the gradients (zero_grad(set_to_none=True)), with a RAGGED batch in the middle of a cycle (the last batch of a
    loader with drop_last=False: first sight of a shape runs eagerly — FasterViT builds its window maps for that batch
    size with a pageable copy, illegal under capture — and an eager micro-batch after replayed ones must ADD to the
    arena slots the replay wrote), for all three model families."""
    from deepfakedetection_amd.graph_step import GraphedTrainStep
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    _, size = _build(family)
    bs = 8 if family != "fastervit" else 4
    g = torch.Generator().manual_seed(3)
    cycles = 6
    sizes = [bs] * (cycles * accum)
    ragged_at = 4 * accum + (1 if accum > 1 else 0)      # a "next" micro-batch (a "first" one when accum == 1) of cycle 5
    sizes[ragged_at] = bs - 2
    batches = [(torch.randn(n, 3, size, size, generator=g).cuda(), torch.randint(0, 2, (n,), generator=g).cuda()) for n in sizes]

    def run(graph: bool):
        torch.manual_seed(11)
        model, _ = _build(family)
        model = model.cuda().train()
        opt = HipAdamW(model.parameters(), lr=1e-3, weight_decay=5e-2)
        step = GraphedTrainStep(model, HipCrossEntropyLoss(0.1), opt, accum_steps=accum)
        if not graph:
            step.failed = True                              # the object's own eager path
        losses = []
        for i, (x, y) in enumerate(batches):
            if i == 3 * accum:
                opt.zero_grad(set_to_none=True)              # what train_one_epoch does at the start of an epoch
            losses.append(step.micro_batch(x, y, first=i % accum == 0, last=(i + 1) % accum == 0).clone())
            if (i + 1) % accum == 0:
                step.optimizer_step()
        torch.cuda.synchronize()
        return model, torch.stack(losses).cpu(), step

    m_e, l_e, _ = run(False)
    m_g, l_g, step = run(True)
    # cycle 1 eager (first sight of every key), cycles 2.. replayed except the ragged micro-batch
    assert not step.failed and step.step_graph is not None
    assert step.replays == (cycles - 1) * accum - 1, step.replays
    assert torch.equal(l_e, l_g), (l_e, l_g)
    for (n1, a), (_, b) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, b), n1
    first = next(iter(m_e.parameters()))
    torch.manual_seed(11)
    fresh, _ = _build(family)
    assert float((first.detach() - next(iter(fresh.parameters())).detach().cuda()).abs().max()) > 0   # it trained


def test_replay_refuses_stale_addresses():
    """A hipGraph records raw addresses (the k_pw_tn fault was a cached buffer replaced behind a graph's
    back).  Every externally owned tensor a capture touched is journalled; train/eval and dtype switches keep their
    caches (one entry per dtype) so replays continue, while a cache that really is rebuilt makes the next replay raise
    StaleGraphError instead of touching freed memory."""
    import gc

    from deepfakedetection_amd.efficientnet import HipEfficientNet
    from deepfakedetection_amd.graph_step import GraphedTrainStep, StaleGraphError
    from deepfakedetection_amd.optim import HipAdamW, HipCrossEntropyLoss

    torch.manual_seed(5)
    model = HipEfficientNet("b0", "timm", 2).cuda().train()
    opt = HipAdamW(model.parameters(), lr=1e-3)
    step = GraphedTrainStep(model, HipCrossEntropyLoss(0.1), opt)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 2, (8,), generator=g).cuda()
    for _ in range(3):
        step.micro_batch(x, y, first=True)
        step.optimizer_step()
    assert step.replays >= 1 and not step.failed
    entry = next(iter(step.graphs.values()))
    guard = entry[-1]
    assert len(guard) > 100                                  # parameters, BN buffers, derived weights, arena, rng, ...
    ptrs = {e[1] for e in guard}
    assert model.conv_stem.weight.data_ptr() in ptrs and opt.arena.flat.data_ptr() in ptrs
    # train -> eval (f32, inference mode) -> train: every recorded tensor is still where it was
    model.eval()
    with torch.inference_mode():
        model(x)
    model.train()
    step.micro_batch(x, y, first=True)
    step.optimizer_step()
    torch.cuda.synchronize()
    # now really rebuild a cache the graph reads: the replay must refuse
    model.__dict__["_derived_caches"].clear()
    gc.collect()
    with pytest.raises(StaleGraphError):
        step.micro_batch(x, y, first=True)
    torch.cuda.synchronize()


@pytest.mark.parametrize("family", ["efficientnet", "efficientformer", "fastervit"])
def test_graphed_eval_forward_is_bitwise_the_eager_forward(family):
    """evaluate() replays its forward per batch shape (graph_step.GraphedForward through trainers.efficientnet.eval_forward):
    same logits bit for bit, also after the weights and BatchNorm statistics changed in between (the graph re-derives
    coefficients and derived weights from the live tensors), with a ragged last batch."""
    from deepfakedetection_amd.graph_step import GraphedForward

    torch.manual_seed(2)
    model, size = _build(family)
    model = model.cuda().eval()
    fwd = GraphedForward(model)
    g = torch.Generator().manual_seed(9)
    bs = 8 if family != "fastervit" else 4
    xs = [torch.randn(n, 3, size, size, generator=g).cuda().contiguous(memory_format=torch.channels_last) for n in (bs, bs, bs, bs - 3, bs)]
    with torch.inference_mode():
        for i, x in enumerate(xs):
            if i == 3:
                with torch.no_grad():
                    for p in model.parameters():
                        p.mul_(1.01)
                    for b in model.buffers():
                        if b.dtype == torch.float32:
                            b.add_(0.01)
            got = fwd(x).clone()
            want = model(x)
            assert torch.equal(got, want), (family, i)
    assert fwd.replays == 3 and not fwd.failed
