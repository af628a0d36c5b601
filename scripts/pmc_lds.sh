#!/bin/bash
# LDS counters per kernel and launch shape:  bash scripts/pmc_lds.sh <tag> <script> [args]   (e.g. dw ef -> scripts/dw_shapes.py ef)
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc_lds_$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT" -o g -- python3 $REPO/scripts/"$@" > "$OUT/run.log" 2> "$OUT/run.err"
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not k.startswith(("void k_", "k_")): continue
        key=(k[:64], r.get("Grid_Size","?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v","?")))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    c=lambda n: sum(v[n])/max(len(v[n]),1) if n in v else 0.0
    idx, conf, gui = c("SQ_LDS_IDX_ACTIVE"), c("SQ_LDS_BANK_CONFLICT"), c("GRBM_GUI_ACTIVE")
    print(f"{k[0]:64s} grid {k[1]:>9s} lds {k[2]:>6s}  idx_active {idx:12.0f} conflict {conf:12.0f} ({conf/max(idx,1):.2f})  lds_busy/gui {idx/max(gui,1)/256*8:.2f}  n={len(v['SQ_LDS_IDX_ACTIVE'])}")
PY
