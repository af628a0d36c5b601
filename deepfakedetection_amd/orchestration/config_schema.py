"""YAML schema of the orchestrator — behaviour-identical mirror of the reference's
orchestration/config_schema.py (defaults :8-55, validators :57-89).

The reference's shipped YAML files must parse to exactly the same plain dicts; that is
pinned by tests/test_contract.py against values captured from the reference module itself
(tests/golden/reference_contract.json).

Unknown keys: dropped inside `data`, kept everywhere else (trainers read bespoke knobs
such as lr / weight_decay / accum_steps from the `training` block).
"""

from __future__ import annotations

from typing import Any, Optional, Union

from pydantic import BaseModel, ConfigDict, Field, field_validator, model_validator


class _Open(BaseModel):
    """Blocks that tolerate and preserve keys this schema does not know."""

    model_config = ConfigDict(extra="allow")


class DataConfig(BaseModel):
    model_config = ConfigDict(extra="ignore")

    root: str = Field(..., description="Dataset root; ImageFolder splits live underneath.")
    train_split: str = "train"
    val_split: str = "val"
    test_split: str = "test"
    num_classes: int = 2
    img_size: int = 224
    class_labels: Optional[dict[str, str]] = None


class TrainingConfig(_Open):
    batch_size: int = 64
    epochs: int = 10
    num_workers: int = 4
    img_size: Optional[int] = None
    transforms: Optional[dict[str, Any]] = None
    resume: Union[str, bool, None] = None


class InferenceConfig(_Open):
    weights: Optional[str] = None
    split: Optional[str] = None
    batch_size: int = 64
    num_workers: int = 4
    img_size: Optional[int] = None
    transforms: Optional[dict[str, Any]] = None


class ModelConfig(_Open):
    output_dir: Optional[str] = None
    transforms: Optional[dict[str, Any]] = None
    training: Optional[TrainingConfig] = None
    inference: Optional[InferenceConfig] = None
    display_name: Optional[str] = None
    label: Optional[str] = None


class OrchestratorConfig(_Open):
    seed: Optional[int] = None
    device: Optional[str] = None
    data: DataConfig
    models: dict[str, ModelConfig]
    selection: Optional[list[str]] = None

    @field_validator("models")
    @classmethod
    def _models_present(cls, models: dict[str, ModelConfig]) -> dict[str, ModelConfig]:
        if len(models) == 0:
            raise ValueError("config.models cannot be empty")
        return models

    @model_validator(mode="after")
    def _resolve_selection(self) -> "OrchestratorConfig":
        known = self.models or {}
        if self.selection is None:
            self.selection = [name for name in known]
        else:
            unknown = [name for name in self.selection if name not in known]
            if unknown:
                raise ValueError("selection references unknown models: " + ", ".join(unknown))
        return self


__all__ = ["DataConfig", "InferenceConfig", "ModelConfig", "OrchestratorConfig", "TrainingConfig"]
