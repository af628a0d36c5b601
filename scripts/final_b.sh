#!/bin/bash
# round-end validation, part B: the full GPU suite again (after the last test change) and the counter passes of the default workload
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
bash scripts/profile_pmc.sh r04 > gpurun_out/final/pmc_b0.log 2>&1; echo "pmc b0 rc=$?"
