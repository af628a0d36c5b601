#!/bin/bash
# SQ counters of ONE kernel of one block under a DFD_TUNE setting: bash scripts/pmc_sq.sh <op> <block> <tag> [DFD_TUNE]
set -e
OP=$1; BLK=$2; TAG=${3:-pmc}; export DFD_TUNE=${4:-}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_INST_CYCLES_VMEM"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT" -o g$i -- python3 $REPO/scripts/run_one.py $OP $BLK 3 > /dev/null 2> "$OUT/g$i.err" || { tail -3 "$OUT/g$i.err"; }
  i=$((i+1))
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not k.startswith(("void k_dw","void k_pw")): continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()): print(f"   {c:32s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
PY
