"""Drop-in for the reference's inference.py: `python inference.py --config config/inference_mi355x.yaml`
(threshold selection on the validation split, metrics.jsonl, plots — through the orchestrator)."""
from deepfakedetection_amd.cli import run

if __name__ == "__main__":
    run("inference", "config/inference_mi355x.yaml")
