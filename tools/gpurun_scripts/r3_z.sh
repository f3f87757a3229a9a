#!/bin/bash
# round 3, z: the attention-related GPU tests under each environment switch that changes the attention path
mkdir -p gpurun_out/r3
for sw in WF_ATTN_PRESCALE=0 WF_ATTN_TRACK_MAX=1 WF_CROSS_FUSED=0 WF_NORM_BOUND_PASS=1; do
  env $sw python -m pytest tests/test_gpu_dit.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py tests/test_gpu_longcat.py tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r3/z_$sw.log 2>&1
  echo "== $sw rc=$? $(grep -E 'passed|failed' gpurun_out/r3/z_$sw.log | tail -1)" | tee -a gpurun_out/r3/z_switches.txt
done
