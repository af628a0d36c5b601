"""`python train.py --config config/train_mi355x.yaml` — training through the orchestrator
(drop-in for the reference's train.py: same flag, same YAML schema)."""

from __future__ import annotations

import argparse
from pathlib import Path

from deepfakedetection_amd.orchestration.orchestrator import orchestrate


def main() -> None:
    cli = argparse.ArgumentParser(description="Train deepfake detectors on the MI355X engine")
    cli.add_argument("--config", type=Path, default=Path("config/train_mi355x.yaml"))
    orchestrate(cli.parse_args().config.resolve(), mode="training")


if __name__ == "__main__":
    main()
