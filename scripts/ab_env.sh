#!/bin/bash
# A/B of one environment switch over bench.py:  bash scripts/ab_env.sh VAR "model args" -> ms/step for VAR=0 and VAR=1
VAR=$1; shift
for v in 0 1; do
  for m in "" "--model efficientformerv2_s1"; do
    env $VAR=$v python bench.py --no-cpu-baseline --extra-models "" --profile-steps 0 --eval-steps 0 $m "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', '$m', d['ms_per_step'], d['value'])"
  done
done
