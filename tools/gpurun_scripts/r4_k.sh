#!/bin/bash
# round 4, k: tools/conv_check.py (hash of wf_conv3d_333 outputs, 3 runs per shape) under the previous conv.hip and the round-4 one
#   -> gpurun_out/r4/k_conv_check.txt
mkdir -p gpurun_out/r4
for v in conv_old NEW; do
  echo "== $v" >> gpurun_out/r4/k_conv_check.txt
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 600 python tools/conv_check.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/k_conv_check.txt
done
cat gpurun_out/r4/k_conv_check.txt
