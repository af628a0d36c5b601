set -e
timeout -k 10 800 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q -k "pwconv or fullsize or full_size or config2" > gpurun_out/t_pw.log 2>&1 || { tail -30 gpurun_out/t_pw.log; exit 1; }
tail -2 gpurun_out/t_pw.log
python scripts/bench_layers.py pw 2>/dev/null | grep -E "pw_project |pw_proj_wgrad|totals|ms  " > gpurun_out/layers_pw2.log
cat gpurun_out/layers_pw2.log | awk '{printf "%s ", $0; if (NR%2==0) print ""}' | cut -c1-200
