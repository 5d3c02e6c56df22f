set -u
export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
timeout 900 python -m pytest -m gpu -q tests/test_gpu_groups.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py::test_mfma_fp16_checkpoint_zero_valued_salients tests/test_gpu_pack.py -k "not fuse_decode" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; tail -5 $O/t1.log
timeout 900 python -m pytest -m gpu -q tests/test_gpu_parity.py -k "mfma or llama13b or g7 or small_batch" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -3 $O/t2.log
PBL_BENCH_SHAPES="13824x5120:0.8,4096x4096:0.9" PBL_BENCH_M=32,16 timeout 600 python tools/bench_mfma.py 2>&1 | tail -1 | tee $O/mfma.json
