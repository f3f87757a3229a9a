#!/bin/bash
# round 4, n: tools/gemm_check.py (hashes of wf_gemm_bf16 on the VAE's own GEMM shapes) under the defective library (lab conv_old) and HEAD
#   -> gpurun_out/r4/n_gemm_check_{conv_old,NEW}.txt
mkdir -p gpurun_out/r4
for v in conv_old NEW; do
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 900 python tools/gemm_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/n_gemm_check_$v.txt
done
diff gpurun_out/r4/n_gemm_check_conv_old.txt gpurun_out/r4/n_gemm_check_NEW.txt; echo "diff rc $?"; cat gpurun_out/r4/n_gemm_check_NEW.txt
