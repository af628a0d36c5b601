set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fastervit_gpu.py tests/test_efformer_gpu.py tests/test_vit_ops_gpu.py -x -q > gpurun_out/t_vit.log 2>&1 || { tail -30 gpurun_out/t_vit.log; exit 1; }
tail -2 gpurun_out/t_vit.log
python bench.py --model faster_vit_0_224 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
python bench.py --model efficientformerv2_s1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330
