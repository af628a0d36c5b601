#!/bin/bash
# round-end measurement on the final sources: default bench line, kernel statistics of the three workloads, counter passes of the default workload
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"
bash scripts/profile_round.sh r04 > gpurun_out/final/profile_round.log 2>&1; echo "profile_round rc=$?"
bash scripts/profile_pmc.sh r04 > gpurun_out/final/pmc_b0.log 2>&1; echo "pmc b0 rc=$?"
