"""Drop-in for the reference's train.py: `python train.py --config config/train_mi355x.yaml`
(same flag, same YAML schema; single node multi-GPU: launch it with torch.distributed.run)."""
from deepfakedetection_amd.cli import run

if __name__ == "__main__":
    run("training", "config/train_mi355x.yaml")
