#!/bin/bash
# after final_e.sh and a fresh `python bench.py > gpurun_out/final/bench_default.json`: write profiles/r04_*
set -e
python scripts/make_profile_summary.py r04 gpurun_out/prof_r04 gpurun_out/pmc_r04 gpurun_out/final/bench_default.json > /dev/null
python scripts/make_profile_summary.py r04 gpurun_out/prof_r04 gpurun_out/pmc_r04ef gpurun_out/final/bench_default.json ef:efficientformerv2_s1 | tail -1 | cut -c1-120
python scripts/make_profile_summary.py r04 gpurun_out/prof_r04 gpurun_out/pmc_r04fv gpurun_out/final/bench_default.json fv:faster_vit_0_224 | tail -1 | cut -c1-120
for k in b0 ef fv; do cp gpurun_out/prof_r04/${k}_table.txt profiles/r04_${k}_kernel_table.txt; done
cp gpurun_out/prof_r04/ef_kernel_stats.csv profiles/r04_ef_kernel_stats.csv
cp gpurun_out/prof_r04/fv_kernel_stats.csv profiles/r04_fv_kernel_stats.csv
cp gpurun_out/final/trainer_b0.jsonl profiles/r04_trainer_loop_b0.jsonl
tail -8 profiles/r04_summary.md | cut -c1-300
