#!/bin/bash
# round 3, t: rare blocks out of line in k_attn_w4 / k_conv_w4 (no taken branch in the common case), four tiles per back edge in the hot
# attention loop: the tests that run those kernels, the conv and attention micro-benches, the bench line
set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py tests/test_gpu_bsa.py tests/test_gpu_longcat.py tests/test_gpu_multirank.py tests/test_gpu_vae.py tests/test_gpu_sampler.py -x -q -m gpu > gpurun_out/r3/t_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3/t_tests.log
grep -n "passed\|failed\|rc=" gpurun_out/r3/t_tests.log
python tools/conv_bench.py > gpurun_out/r3/t_conv_bench.txt 2>&1
tail -12 gpurun_out/r3/t_conv_bench.txt
python tools/cross_attn_bench.py 2>/dev/null | grep "L=" > gpurun_out/r3/t_cross_attn.txt
cat gpurun_out/r3/t_cross_attn.txt
python bench.py --no-cpu-baseline > gpurun_out/r3/t_bench.json 2> gpurun_out/r3/t_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3/t_bench.json'))
print(d['value'], d['guided_step_ms'], d['plain_step_ms'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['tracked_body_frac'])
PY
