#!/bin/bash
# rocprofv3 kernel statistics of ONE bench workload:  bash scripts/profile_model.sh <tag> [bench.py args...]
#   -> gpurun_out/prof_<tag>/{kernel_stats.csv,table.txt,line.json}
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-steps 0 --extra-models "" "$@" > "$OUT/line.json" 2> "$OUT/run.err"
find "$OUT" -name "*kernel_trace.csv" -delete
cd "$REPO"
f=$(find "$OUT" -name "run_kernel_stats.csv" | head -1)
python3 scripts/kstats.py "$f" 60 > "$OUT/table.txt"
cp "$f" "$OUT/kernel_stats.csv"
