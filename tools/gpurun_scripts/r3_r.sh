#!/bin/bash
# round 3, r: after the two instruction-count cuts in k_attn_w4 (row-sum fold, one-instruction M0): every test that runs an attention kernel, then the bench
set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py tests/test_gpu_bsa.py tests/test_gpu_longcat.py tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r3/r_attn_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3/r_attn_tests.log
tail -5 gpurun_out/r3/r_attn_tests.log
python tools/attn_bench.py > gpurun_out/r3/r_attn_bench.txt 2>&1
tail -8 gpurun_out/r3/r_attn_bench.txt
python bench.py --no-cpu-baseline > gpurun_out/r3/r_bench.json 2> gpurun_out/r3/r_bench.err
tail -c 1500 gpurun_out/r3/r_bench.json
