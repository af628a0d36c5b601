timeout -k 10 300 python -m pytest tests/test_vit_ops_gpu.py -m gpu -x -q -k "attn_softmax or attention_products" 2>&1 | tail -4
timeout -k 10 400 python -m pytest tests/test_efformer_gpu.py tests/test_fullsize_vit_gpu.py tests/test_fastervit_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --model efficientformerv2_s1 --no-cpu-baseline --profile-steps 0 --eval-steps 0 --extra-models none 2>/dev/null | python3 scripts/bench_ms.py
