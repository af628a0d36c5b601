#!/bin/bash
# ms per step of a bench workload, N runs: bash scripts/bench_ms.sh [model] [runs]
M=${1:-}; N=${2:-2}
for i in $(seq $N); do
  python bench.py ${M:+--model $M} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${M:-b0}', d['value'], d['ms_per_step'])"
done
