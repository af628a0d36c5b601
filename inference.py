"""`python inference.py --config config/inference_mi355x.yaml` — batch evaluation through
the orchestrator (drop-in for the reference's inference.py)."""

from __future__ import annotations

import argparse
from pathlib import Path

from deepfakedetection_amd.orchestration.orchestrator import orchestrate


def main() -> None:
    cli = argparse.ArgumentParser(description="Evaluate deepfake detectors on the MI355X engine")
    cli.add_argument("--config", type=Path, default=Path("config/inference_mi355x.yaml"))
    orchestrate(cli.parse_args().config.resolve(), mode="inference")


if __name__ == "__main__":
    main()
