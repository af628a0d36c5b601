#!/bin/bash
# rocprofv3 kernel statistics of the three bench workloads (run on the GPU box from the repo root):
#   bash scripts/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{b0,ef,fv}_kernel_stats.csv
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
COMMON="--steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-steps 0 --extra-models none"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o b0 -- python3 $REPO/bench.py $COMMON > "$OUT/b0.json" 2> "$OUT/b0.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ef -- python3 $REPO/bench.py --model efficientformerv2_s1 $COMMON > "$OUT/ef.json" 2> "$OUT/ef.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o fv -- python3 $REPO/bench.py --model faster_vit_0_224 $COMMON > "$OUT/fv.json" 2> "$OUT/fv.err"
find "$OUT" -name "*kernel_trace.csv" -delete          # hundreds of MB; the stats are what is kept
cd "$REPO"
for k in b0 ef fv; do
  f=$(find "$OUT" -name "${k}_kernel_stats.csv" | head -1)
  python3 scripts/kstats.py "$f" 45 > "$OUT/${k}_table.txt"
done
