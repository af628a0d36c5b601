#!/bin/bash
# rocprofv3 kernel statistics of the default bench (B0): bash scripts/profile_b0.sh <tag> -> gpurun_out/prof_<tag>/b0_table.txt
set -e
TAG=${1:-x}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o b0 -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-steps 0 ${@:2} > "$OUT/b0.json" 2> "$OUT/b0.err"
find "$OUT" -name "*kernel_trace.csv" -delete
cd "$REPO"
python3 scripts/kstats.py "$(find "$OUT" -name "b0_kernel_stats.csv" | head -1)" 70 > "$OUT/b0_table.txt"
