#!/bin/bash
# Timing-only ablations of the depthwise forward kernel (results are WRONG by construction):
# builds private copies of libdfd_hip.so with -DDFD_ABL=<n> and times block <blk> of EfficientNet-B0.
set -e
cd "$(dirname "$0")/.."
for abl in "$@"; do
  echo "== DFD_ABL=$abl"
  DFD_EXTRA_FLAGS="-DDFD_ABL=$abl" python -m deepfakedetection_amd.build --force > /dev/null 2>&1
  for blk in 2 4 9; do python scripts/run_one.py dw_fwd $blk 20; done
done
python -m deepfakedetection_amd.build --force > /dev/null 2>&1
