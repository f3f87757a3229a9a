#!/bin/bash
# round 3, s: attention epilogue stores as 8 x dwordx4 per lane (was 16 x dwordx2): previous library vs new, same box; then every attention test
set -x
mkdir -p gpurun_out/r3
out=gpurun_out/r3/s_epilogue_ab.txt
: > $out
for round in 1 2; do
  for lib in worldforge_amd/_lib/lab/libwf_hip_prev.so worldforge_amd/_lib/libwf_hip.so; do
    echo "== $lib (round $round)" >> $out
    WF_LIB=$lib python tools/cross_attn_bench.py 2>/dev/null | grep "L=" >> $out
    WF_LIB=$lib python tools/attn_bench.py 2>/dev/null | grep "L=32760 pre-scaled Q + key-norm" | tail -1 >> $out
  done
done
cat $out
python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py tests/test_gpu_bsa.py tests/test_gpu_longcat.py tests/test_gpu_multirank.py tests/test_gpu_sampler.py tests/test_gpu_longcat_sampler.py -x -q -m gpu > gpurun_out/r3/s_attn_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3/s_attn_tests.log
grep -n "passed\|failed\|rc=" gpurun_out/r3/s_attn_tests.log
