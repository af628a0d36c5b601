#!/bin/bash
# Counters of the NT / TN GEMM kernels on FasterViT's Linear shapes:  bash scripts/pmc_gemm.sh  -> gpurun_out/pmc_gemm/summary.txt
set -e
OUT=$PWD/gpurun_out/pmc_gemm
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT" -o g$i -- python3 $REPO/scripts/gemm_shapes.py > "$OUT/g$i.log" 2> "$OUT/g$i.err" || echo "group $i failed"
  i=$((i+1))
done
cd "$REPO"
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if not ("k_pw_nt" in k or "k_pw_tn" in k): continue
        key=(k[:58], r.get("Grid_Size","?"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    c=lambda n: sum(v[n])/max(len(v[n]),1) if n in v else float("nan")
    gui=c("GRBM_GUI_ACTIVE")/8
    print(f"{k[0]:58s} grid {k[1]:>8s} cyc {gui:9.0f} parked {c('SQ_WAIT_ANY')/max(c('SQ_WAVE_CYCLES'),1):.2f} wait_lds/wavecyc {c('SQ_WAIT_INST_LDS')/max(c('SQ_WAVE_CYCLES'),1):.2f} "
          f"lds_busy {c('SQ_LDS_IDX_ACTIVE')/max(gui*256,1):.2f} conflict {c('SQ_LDS_BANK_CONFLICT')/max(c('SQ_LDS_IDX_ACTIVE'),1):.2f} "
          f"mfma_busy {c('SQ_VALU_MFMA_BUSY_CYCLES')/max(c('SQ_BUSY_CYCLES'),1):.3f} tcc_hit {c('TCC_HIT_sum')/max(c('TCC_HIT_sum')+c('TCC_MISS_sum'),1):.2f} "
          f"tcc_req {c('TCC_REQ_sum'):.3g} ea_rd {c('TCC_EA0_RDREQ_sum'):.3g} tcp_rd {c('TCP_TCC_READ_REQ_sum'):.3g}")
PY
cat "$OUT/summary.txt"
