#!/bin/bash
# rocprofv3 kernel statistics of one bench workload: bash scripts/profile_model.sh <tag> <model> -> gpurun_out/prof_<tag>/table.txt
set -e
TAG=$1; MODEL=$2
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o m -- python3 $REPO/bench.py --model $MODEL --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-steps 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT" -name "*kernel_trace.csv" -delete
cd "$REPO"
python3 scripts/kstats.py "$(find "$OUT" -name "m_kernel_stats.csv" | head -1)" 60 > "$OUT/table.txt"
