set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -x --durations=12 > gpurun_out/s4/configs.log 2>&1; echo "rc=$?" >> gpurun_out/s4/configs.log; tail -25 gpurun_out/s4/configs.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mfma or g6 or compile or grouped" > gpurun_out/s4/parity_sub.log 2>&1; echo "rc=$?" >> gpurun_out/s4/parity_sub.log; tail -5 gpurun_out/s4/parity_sub.log
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" timeout 300 python tools/bench_mfma.py > gpurun_out/s4/mfma.json 2>&1; tail -2 gpurun_out/s4/mfma.json
timeout 600 python tools/bench_llama7b.py > gpurun_out/s4/llama7b.json 2> gpurun_out/s4/llama7b.err; tail -3 gpurun_out/s4/llama7b.json; tail -5 gpurun_out/s4/llama7b.err
