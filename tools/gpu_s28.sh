set -u
export TMPDIR=/tmp
O=gpurun_out/s28; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_gpu_configs.py tests/test_gpu_groups.py -k "fused or graph or decode" 2>&1 | grep -E "^E  |passed|failed" | head
timeout 600 python tools/bench_prefetch.py --gap 4 2>/dev/null | tail -1 | tee $O/decode_gap4.json
timeout 600 python tools/bench_decoder_layer.py 2>/dev/null | head -1 | tee $O/decoder_layer.json
