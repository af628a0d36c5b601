set -e
timeout -k 10 900 python -m pytest tests/test_vit_ops_gpu.py tests/test_fastervit_gpu.py tests/test_efformer_gpu.py -x -q > gpurun_out/t_vit.log 2>&1 || { tail -30 gpurun_out/t_vit.log; exit 1; }
tail -1 gpurun_out/t_vit.log
for m in faster_vit_0_224 efficientformerv2_s1; do
python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b_$m.json
python -c "import json,sys; d=json.load(open('gpurun_out/b_$m.json')); print('$m', d['value'], d['ms_per_step'])"
done
