"""The plug-in surface (YAML schema, env-var contract, registry, checkpoints) against values
captured from the REFERENCE's own modules (tests/golden/reference_contract.json, written by
tests/golden/make_golden.py importing /root/reference/orchestration/{config_schema,train_env}.py)
and against the contract table of SURVEY.md App. A.  CPU only.
"""

from __future__ import annotations

import json
import os
from pathlib import Path

import pytest
import torch
from pydantic import ValidationError

from deepfakedetection_amd.orchestration import model_registry as reg
from deepfakedetection_amd.orchestration import orchestrator as orch
from deepfakedetection_amd.orchestration import train_env as te
from deepfakedetection_amd.orchestration.config_schema import OrchestratorConfig

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_contract.json").read_text())


@pytest.mark.parametrize("name", ["train.yaml", "train_imagenette.yaml", "inference.yaml"])
def test_reference_yaml_parses_to_the_reference_dict(name):
    assert OrchestratorConfig(**GOLD["config_inputs"][name]).model_dump() == GOLD["configs"][name]


def test_defaults_and_extra_key_policy():
    assert OrchestratorConfig(data={"root": "d"}, models={"a": {}}).model_dump() == GOLD["defaults"]
    assert OrchestratorConfig(**GOLD["extras_input"]).model_dump() == GOLD["extras_output"]


@pytest.mark.parametrize("case", ["empty_models", "unknown_selection", "missing_data"])
def test_validation_errors_match(case):
    with pytest.raises(ValidationError) as err:
        OrchestratorConfig(**GOLD["validation_inputs"][case])
    assert [e["msg"] for e in err.value.errors()] == GOLD["validation_errors"][case]


def test_env_readers_and_toggles(monkeypatch):
    env = GOLD["env"]
    monkeypatch.setenv("TRANSFORMS", '{"train_to_tensor":false,"train_color_jitter":"yes","x":0}')
    got = te.load_transform_toggles({"train_to_tensor": True, "train_color_jitter": False}, required=("train_to_tensor",))
    assert got == env["toggles"]
    monkeypatch.setenv("TRANSFORMS", "not json")
    assert te.load_transform_toggles({"a": True, "b": False}) == env["toggles_bad_json"]
    monkeypatch.delenv("TRANSFORMS")
    for value in ("abc", "3", " 4", "5.0"):
        monkeypatch.setenv("EPOCHS", value)
        assert te.env_int("EPOCHS", 7) == env[f"env_int[{value}]"]
        assert te.env_float("EPOCHS", 7.5) == env[f"env_float[{value}]"]
    monkeypatch.delenv("EPOCHS")
    assert te.env_int("EPOCHS", 7) == env["env_int[unset]"]
    monkeypatch.delenv("TRAIN_SPLIT", raising=False)
    assert te.env_str("TRAIN_SPLIT", "Train") == env["env_str[unset]"]
    for v in (True, False, 1, 0, 2.5, "1", "true", " YES ", "on", "off", "", None, [1]):
        assert te.as_bool(v) == env["as_bool"][repr(v)], v
    assert sorted(te.__all__) == env["__all__"]


def test_require_num_classes_messages():
    class Fake:
        classes = ["a", "b", "c", "d", "e", "f", "g"]

    with pytest.raises(ValueError) as err:
        te.require_num_classes(Fake(), 2, split="train", dataset_root="/data/x")
    assert str(err.value) == GOLD["env"]["require_num_classes_error"]
    with pytest.raises(ValueError) as err:
        te.require_num_classes(Fake(), 0, split="train")
    assert str(err.value) == GOLD["env"]["require_num_classes_nonpositive"]
    te.require_num_classes(Fake(), 7, split="train")
    te.require_num_classes(object(), 3, split="train")           # no .classes -> accepted


def test_prepare_environment_and_resume(tmp_path, monkeypatch):
    monkeypatch.setenv("OUTPUT_DIR", str(tmp_path / "run"))
    monkeypatch.setenv("SEED", "5")
    monkeypatch.setenv("DEVICE", "cuda:1")
    monkeypatch.setenv("RESUME_AUTO", "1")
    env = te.prepare_training_environment(weights_name="W.pth")
    assert env.checkpoints_dir.is_dir() and env.logs_dir.is_dir()
    assert env.best_weights_path == (tmp_path / "run" / "W.pth").resolve()
    assert env.latest_checkpoint_path.name == "latest.ckpt" and env.best_checkpoint_path.name == "best.ckpt"
    assert (env.seed, env.device_override, env.resume_checkpoint) == (5, "cuda:1", None)   # no latest.ckpt yet
    model = torch.nn.Linear(3, 2)
    opt = torch.optim.AdamW(model.parameters())
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=3)
    state = te.save_latest_checkpoint(env, model=model, optimizer=opt, scheduler=sched, epoch=2, best_val_acc=0.5,
                                      best_epoch=1, extra={"warmup_done": True})
    assert set(state) == {"epoch", "model", "optimizer", "scheduler", "best_val_acc", "best_epoch", "warmup_done"}
    te.save_best_checkpoint(env, state)
    assert set(torch.load(env.best_weights_path)) == {"weight", "bias"}               # bare state_dict
    env2 = te.prepare_training_environment(weights_name="W.pth")
    assert env2.resume_checkpoint == env.latest_checkpoint_path
    other = torch.nn.Linear(3, 2)
    loaded = te.maybe_load_checkpoint(env2, model=other, optimizer=opt, scheduler=sched)
    assert loaded["epoch"] == 2 and torch.equal(other.weight, model.weight)
    monkeypatch.setenv("RESUME_AUTO", "0")
    assert te.prepare_training_environment(weights_name="W.pth").resume_checkpoint is None


def _run_paths(tmp_path):
    return orch.ensure_run_dirs(tmp_path / "runs" / "m", "20250101-000000")


def test_build_env_overrides_training_table(tmp_path):
    """SURVEY.md App. A (reference orchestrator.py:183-283)."""
    cfg = OrchestratorConfig(**GOLD["config_inputs"]["train_imagenette.yaml"]).model_dump()
    rp = _run_paths(tmp_path)
    m = {"name": "efficientformerv2_s1", **cfg["models"]["efficientformerv2_s1"]}
    env = orch.build_env_overrides(config=cfg, model_cfg=m, run_paths=rp, training=True)
    want = {"OUTPUT_DIR": str(rp.run_dir), "SEED": "1", "DEVICE": "cuda", "TRAIN_SPLIT": "train", "VAL_SPLIT": "val",
            "TEST_SPLIT": "test", "NUM_CLASSES": "10", "BATCH_SIZE": "192", "EPOCHS": "2", "NUM_WORKERS": "4",
            "IMG_SIZE": "224", "RESUME_AUTO": "0", "DATA_ROOT": str(Path("data/imagenette2-160").resolve())}
    for key, val in want.items():
        assert env[key] == val, key
    assert json.loads(env["TRANSFORMS"]) == cfg["models"]["efficientformerv2_s1"]["transforms"]["train"]
    assert "LR" not in env and "ACCUM_STEPS" not in env and "LOG_PATH" not in env
    b3 = {"name": "efficientnet_b3", **cfg["models"]["efficientnet_b3"]}
    env = orch.build_env_overrides(config=cfg, model_cfg=b3, run_paths=rp, training=True)
    assert env["IMG_SIZE"] == "160" and env["RESUME_AUTO"] == "1" and env["MODEL_NAME"] == "efficientnet_b3"
    b3["training"] = {**b3["training"], "lr": 0.01, "weight_decay": 0.1, "accum_steps": 2, "warmup_epochs": 3,
                      "early_stop_patience": 9, "resume": True}
    env = orch.build_env_overrides(config=cfg, model_cfg=b3, run_paths=rp, training=True)
    assert (env["LR"], env["WEIGHT_DECAY"], env["ACCUM_STEPS"], env["WARMUP_EPOCHS"], env["EARLY_STOP_PATIENCE"],
            env["RESUME_AUTO"]) == ("0.01", "0.1", "2", "3", "9", "1")


def test_build_env_overrides_inference_table(tmp_path):
    cfg = OrchestratorConfig(**GOLD["config_inputs"]["inference.yaml"]).model_dump()
    rp = _run_paths(tmp_path)
    name = "efficientnet_b3"
    env = orch.build_env_overrides(config=cfg, model_cfg={"name": name, **cfg["models"][name]}, run_paths=rp, training=False)
    assert (env["TEST_SPLIT"], env["BATCH_SIZE"], env["NUM_WORKERS"], env["IMG_SIZE"]) == ("test", "256", "2", "224")
    assert "RESUME_AUTO" not in env and "EPOCHS" not in env
    bare = {"name": name, "inference": None, "training": {"batch_size": 8}}
    env = orch.build_env_overrides(config=cfg, model_cfg=bare, run_paths=rp, training=False)
    assert (env["BATCH_SIZE"], env["NUM_WORKERS"], env["IMG_SIZE"]) == ("8", "0", str(cfg["data"]["img_size"]))


def test_transform_mapping_resolution():
    nested = {"transforms": {"train": {"a": True}, "eval": {"b": False}}}
    assert orch.resolve_transform_mapping(nested, phase="train") == {"a": True}
    assert orch.resolve_transform_mapping(nested, phase="eval") == {"b": False}
    flat = {"transforms": {"a": 1, "b": "no"}}
    assert orch.resolve_transform_mapping(flat, phase="eval") == {"a": 1, "b": "no"}
    scoped = {"transforms": None, "training": {"transforms": {"c": True}}, "inference": {}}
    assert orch.resolve_transform_mapping(scoped, phase="train") == {"c": True}
    assert orch.resolve_transform_mapping(scoped, phase="eval") is None


def test_patched_environ_restores(monkeypatch):
    monkeypatch.setenv("KEEP", "old")
    monkeypatch.delenv("FRESH", raising=False)
    with pytest.raises(RuntimeError):
        with orch.patched_environ({"KEEP": "new", "FRESH": "1"}):
            assert os.environ["KEEP"] == "new" and os.environ["FRESH"] == "1"
            raise RuntimeError
    assert os.environ["KEEP"] == "old" and "FRESH" not in os.environ


def test_registry_lookup_semantics():
    spec = reg.get_model_spec("efficientnet_b3")
    assert (spec.name, spec.weights_key, spec.default_image_size) == ("efficientnet_b3", "efficientnet_b3", 224)
    assert spec.train_module.endswith("trainers.efficientnet")
    assert reg.get_model_spec("efficientnet_b0").train_module == spec.train_module
    pre = reg.get_model_spec("faster_vit_0_224")
    assert (pre.name, pre.weights_key) == ("faster_vit_0_224", "faster_vit_0_224")
    assert reg.get_model_spec("efficientformerv2_s1").train_module.endswith("trainers.efficientformer_v2")
    with pytest.raises(KeyError) as err:
        reg.get_model_spec("resnet50")
    assert err.value.args[0] == "Unknown model 'resnet50'. Add it to model_registry.py."
    fv = pre.builder("faster_vit_0_224", 2)
    assert fv.head.in_features == 512 and fv.head.out_features == 2           # trainers/fastervit.py:372-373 swaps model.head
    assert reg.get_model_spec("faster_vit_2_224").builder("faster_vit_2_224", 2).head.in_features == 768
    with pytest.raises(NotImplementedError):
        reg.get_model_spec("faster_vit_4_21k_224").builder("faster_vit_4_21k_224", 2)   # registered prefix, no engine: loud


def test_builders_return_modules_with_reference_surface():
    b3 = reg.get_model_spec("efficientnet_b3").builder("efficientnet_b3", 2)
    assert b3._fc.in_features == 1536 and b3._fc.out_features == 2            # trainers/efficientnet.py:406-407
    assert isinstance(b3._conv_head, torch.nn.Conv2d)                          # web_ui.py:111
    assert any(isinstance(m, torch.nn.Conv2d) for m in b3.modules())
    assert [n for n, _ in b3.named_parameters() if "_fc" in n] == ["_fc.weight", "_fc.bias"]
    b3._fc = torch.nn.Linear(b3._fc.in_features, 5)                           # the reference swaps the head after construction
    assert b3.state_dict()["_fc.weight"].shape == (5, 1536)
    b0 = reg.get_model_spec("efficientnet_b0").builder("efficientnet_b0", 10)
    assert b0.classifier.in_features == 1280 and sum(p.numel() for p in b0.parameters()) == 4_020_358


def test_import_trainer_contract():
    assert callable(orch.import_trainer("deepfakedetection_amd.trainers.efficientnet"))
    with pytest.raises(AttributeError) as err:
        orch.import_trainer("deepfakedetection_amd.arch")
    assert str(err.value) == "Trainer module 'deepfakedetection_amd.arch' must expose a main() function."


def test_threshold_sweep_matches_sklearn_loop():
    import numpy as np
    from sklearn.metrics import balanced_accuracy_score

    rng = np.random.default_rng(0)
    truth = rng.integers(0, 2, 400)
    scores = np.clip(truth * 0.3 + rng.random(400) * 0.7, 0, 1)
    best, chosen = -1.0, 0.5
    for thr in np.linspace(0.0, 1.0, 501, dtype=np.float64):          # the reference's loop (orchestrator.py:533-544)
        bal = balanced_accuracy_score(truth, (scores >= thr).astype(np.int64))
        if bal > best:
            best, chosen = float(bal), float(thr)
    assert orch.best_balanced_accuracy_threshold(scores, truth) == chosen
