#!/bin/bash
# round-end validation: the full GPU suite, then the re-measurement of final_e.sh
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
bash scripts/final_e.sh
