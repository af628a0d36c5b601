for f in 0 1.5 4; do echo "== spin factor $f"; DFD_SPIN_FACTOR=$f python bench.py --no-cpu-baseline --eval-steps 0 --extra-models none 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; print(d['ms_per_step'], r['frac'], r['avg_launch_us'], [(k['kernel'],k['ms_per_step']) for k in d['kernels'][:8]])
"; done
