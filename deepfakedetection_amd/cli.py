"""Command line of the two entry scripts: `--config <yaml>` and nothing else (the flag the reference's
train.py / inference.py take), resolved to an absolute path and handed to the orchestrator."""

from __future__ import annotations

import sys
from argparse import ArgumentParser
from pathlib import Path


def run(mode: str, default_config: str, argv: list[str] | None = None) -> None:
    from .orchestration.orchestrator import orchestrate

    what = {"training": "Train", "inference": "Evaluate"}[mode]
    ap = ArgumentParser(prog=Path(sys.argv[0]).name, description=f"{what} the configured classifiers on the MI355X engine")
    ap.add_argument("--config", type=Path, default=Path(default_config), help="orchestrator YAML (reference schema)")
    orchestrate(ap.parse_args(argv).config.resolve(), mode=mode)
