#!/bin/bash
# re-measurement after a kernel change: default bench line, kernel statistics of the three workloads, counter passes of all three
mkdir -p gpurun_out/final
bash scripts/profile_round.sh r04 > gpurun_out/final/profile_round.log 2>&1; echo "profile_round rc=$?"
bash scripts/profile_pmc.sh r04 > gpurun_out/final/pmc_b0.log 2>&1; echo "pmc b0 rc=$?"
bash scripts/profile_pmc.sh r04ef --model efficientformerv2_s1 > gpurun_out/final/pmc_ef.log 2>&1; echo "pmc ef rc=$?"
bash scripts/profile_pmc.sh r04fv --model faster_vit_0_224 > gpurun_out/final/pmc_fv.log 2>&1; echo "pmc fv rc=$?"
python scripts/bench_trainer.py --steps 120 > gpurun_out/final/trainer_b0.jsonl 2> gpurun_out/final/trainer_b0.err; echo "trainer rc=$?"
