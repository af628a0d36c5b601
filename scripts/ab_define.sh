#!/bin/bash
# A/B of one compile-time constant on the GPU box:  bash scripts/ab_define.sh <csrc file> <MACRO> <v1> <v2> ...
#   rewrites "#define MACRO <default>" in the file, rebuilds, runs the three benches per value; restores the file afterwards
F=deepfakedetection_amd/csrc/$1; M=$2; shift 2
cp "$F" /tmp/ab_define.orig
for v in "$@"; do
  sed -i "s|^#define $M .*|#define $M $v|" "$F"
  python -c "from deepfakedetection_amd import build; build.build()" || exit 1
  for m in efficientnet efficientformerv2_s1 faster_vit_0_224; do
    python bench.py --no-cpu-baseline --extra-models none --profile-steps 0 --eval-steps 0 --model $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M=$v', '$m', d['ms_per_step'], d['value'], flush=True)"
  done
done
cp /tmp/ab_define.orig "$F"
