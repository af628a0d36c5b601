#!/bin/bash
# counter passes of the other two models of the bench line, and the drop-in trainer loop's throughput
mkdir -p gpurun_out/final
bash scripts/profile_pmc.sh r04ef --model efficientformerv2_s1 > gpurun_out/final/pmc_ef.log 2>&1; echo "pmc ef rc=$?"
bash scripts/profile_pmc.sh r04fv --model faster_vit_0_224 > gpurun_out/final/pmc_fv.log 2>&1; echo "pmc fv rc=$?"
python scripts/bench_trainer.py --steps 120 > gpurun_out/final/trainer_b0.jsonl 2> gpurun_out/final/trainer_b0.err; echo "trainer rc=$?"
