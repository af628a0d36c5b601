#!/bin/bash
# B0 inference forward, training-form chain against the inference form, over the size rule's threshold:
#   bash scripts/eval_forms_sweep.sh  ->  gpurun_out/eval_sweep.log   (summarised in profiles/r02_eval_forms.md)
set -e
mkdir -p gpurun_out
for DT in f32 bf16; do
for TH in 0 2097152 8388608 33554432 134217728 99999999999; do
  echo "== $DT max_elems $TH"
  DFD_EVAL_FUSED_MAX_ELEMS=$TH timeout -k 10 200 python scripts/eval_small.py $DT 2>/dev/null
done; done > gpurun_out/eval_sweep.log 2>&1
cat gpurun_out/eval_sweep.log
