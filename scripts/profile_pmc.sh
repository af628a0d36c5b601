#!/bin/bash
# The round's counter passes over the default bench workload (EfficientNet-B0, batch 256, eager launches so that every
# kernel is its own dispatch).  One counter group per process, as the profiling guide prescribes:
#   bash scripts/profile_pmc.sh <tag> [bench.py args, e.g. --model faster_vit_0_224]
#       -> gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE,TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum,SQ}_counter_collection.csv
set -e
TAG=${1:-r02}
shift || true
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 --eval-steps 0 --extra-models none $*"
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT" -o $name -- python3 $REPO/bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "pass $name done"; }
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
run TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run SQ SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
python3 -c "import sys; sys.path.insert(0, '$REPO'); from deepfakedetection_amd.build import source_digest; print(source_digest())" > "$OUT/csrc_sha256.txt"
find "$OUT" -name "*agent_info.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete          # tens of MB per pass; the counter files carry the timestamps
ls -la "$OUT"
