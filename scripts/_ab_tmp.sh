timeout -k 10 300 python -m pytest tests/test_vit_ops_gpu.py -m gpu -x -q -k "attention_products" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_s8 && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o ef -- python3 $GRAFT_REPO_ROOT/bench.py --model efficientformerv2_s1 --steps 4 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-steps 0 --extra-models none > $O/ef.json 2> $O/ef.err
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv,glob
f=glob.glob("gpurun_out/prof_s8/**/ef_kernel_trace.csv",recursive=True)[0]
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Grid_Size_X") or r.get("Grid_Size"),r.get("VGPR_Count","")) for r in csv.DictReader(open(f))))
ad=[i for i,r in enumerate(rows) if "k_adamw" in r[2]]
a,b=ad[-2],ad[-1]
n=0
for s,e,nm,g,v in rows[a+1:b+1]:
    if any(k in nm for k in ("bgemm<","attn_scores","attn_apply")):
        n+=1
        if n<=24: print(f"{(e-s)/1e3:7.1f} us grid {g} vgpr {v} {nm[:44]}")
P
rm -f $(find gpurun_out/prof_s8 -name "*kernel_trace.csv")
for v in 1 0; do echo "== DFD_ATTN_MFMA=$v"; DFD_ATTN_MFMA=$v python bench.py --model efficientformerv2_s1 --no-cpu-baseline --profile-steps 0 --eval-steps 0 --extra-models none 2>/dev/null | python3 scripts/bench_ms.py; done
