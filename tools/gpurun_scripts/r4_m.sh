#!/bin/bash
# round 4, m: the k_gemm_pp dummy-target defect (NI = 2: dummy LDS-DMA region inside wave 7's epilogue staging): tests/test_gpu_gemm_streams.py
# under the defective library (lab library conv_old = HEAD~ gemm.hip) and under the fixed one, 3 times each; then the row-sharded VAE test 3 times
#   -> gpurun_out/r4/m_*.log
mkdir -p gpurun_out/r4
for v in conv_old NEW; do
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  for i in 1 2 3; do
    timeout 900 python -m pytest tests/test_gpu_gemm_streams.py -m gpu -q 2>&1 | grep -E "passed|failed|rror|concurrent runs" | tail -12 >> gpurun_out/r4/m_streams_$v.log
  done
  echo "== $v"; cat gpurun_out/r4/m_streams_$v.log
done
export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_conv_old.so
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "vae" 2>&1 | grep -E "passed|failed" | tail -2 >> gpurun_out/r4/m_vae_sharded_old.log
done
cat gpurun_out/r4/m_vae_sharded_old.log
