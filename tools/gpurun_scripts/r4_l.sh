#!/bin/bash
# round 4, l: which library change broke test_vae_c2_row_sharded_equals_unsharded -- the test under the previous conv.hip (lab library conv_old,
# everything else at HEAD) and under the round-4 conv   -> gpurun_out/r4/l_*.log
mkdir -p gpurun_out/r4
for v in conv_old NEW; do
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -q -k "vae" 2>&1 | grep -E "passed|failed|rror|assert" | tail -8 > gpurun_out/r4/l_$v.log
  echo "== $v"; cat gpurun_out/r4/l_$v.log
done
