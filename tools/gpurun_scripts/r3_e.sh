set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_warp.py tests/test_gpu_thirdparty_goldens.py tests/test_infer_entry.py -m gpu -q --durations=5 > gpurun_out/r3/e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/e_tests.log
tail -25 gpurun_out/r3/e_tests.log
python tools/cross_attn_bench.py > gpurun_out/r3/e_cross_attn_bench.txt 2>&1; cat gpurun_out/r3/e_cross_attn_bench.txt
