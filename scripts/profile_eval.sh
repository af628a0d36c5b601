#!/bin/bash
# rocprofv3 kernel statistics of the B0 inference forward, both kernel chains and both dtypes:
#   bash scripts/profile_eval.sh  ->  gpurun_out/prof_eval_<form>_<dtype>/table.txt
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for FORM in 1 0; do
  for DT in f32 bf16; do
    OUT=$REPO/gpurun_out/prof_eval_fused${FORM}_$DT
    mkdir -p "$OUT"
    export DFD_EVAL_FUSED=$FORM
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o m -- python3 $REPO/scripts/run_eval.py 10 $DT > "$OUT/run.log" 2> "$OUT/run.err"
    find "$OUT" -name "*kernel_trace.csv" -delete
    python3 $REPO/scripts/kstats.py "$(find "$OUT" -name "m_kernel_stats.csv" | head -1)" 40 13 > "$OUT/table.txt"
    echo "fused=$FORM $DT: $(cat $OUT/run.log | tail -1)"
  done
done
