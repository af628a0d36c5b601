#!/bin/bash
# A/B of one environment switch over bench.py:  bash scripts/ab_env.sh VAR [model ...] -> ms/step for VAR=0 and VAR=1
VAR=$1; shift
MODELS=${@:-efficientnet efficientformerv2_s1 faster_vit_0_224}
for v in 0 1; do
  for m in $MODELS; do
    env $VAR=$v python bench.py --no-cpu-baseline --extra-models none --profile-steps 0 --eval-steps 0 --model $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', '$m', d['ms_per_step'], d['value'], flush=True)"
  done
done
