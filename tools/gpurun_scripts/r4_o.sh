#!/bin/bash
# round 4, o: which source breaks test_vae_c2_row_sharded_equals_unsharded -- the 2 x 2 of {conv.hip, gemm.hip} x {HEAD, working tree} as lab
# libraries (tools/lab_lib.py), the test alone and behind tests/test_gpu_vae.py, twice each   -> gpurun_out/r4/o_matrix.txt
mkdir -p gpurun_out/r4
for v in convold_gemmold convold_gemmfix convnew_gemmold NEW; do
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  for i in 1 2; do
    a=$(timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "vae" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$v alone: $a" >> gpurun_out/r4/o_matrix.txt
  done
  a=$(timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py -m gpu -q -k "vae" 2>&1 | grep -E "passed|failed" | tail -1)
  echo "$v behind test_gpu_vae.py: $a" >> gpurun_out/r4/o_matrix.txt
done
cat gpurun_out/r4/o_matrix.txt
