#!/bin/bash
# round-end validation, part A: GPU test suite, the default bench line, rocprofv3 kernel statistics of the three workloads
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest.log
tail -3 gpurun_out/final/pytest.log
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"
bash scripts/profile_round.sh r04 > gpurun_out/final/profile_round.log 2>&1; echo "profile_round rc=$?"
