set -x
mkdir -p gpurun_out/r3
bash tools/gpurun_scripts/conv_pmc.sh > gpurun_out/r3/p_conv_pmc.log 2>&1; tail -25 gpurun_out/r3/p_conv_pmc.log
for sw in "WF_CROSS_FUSED=0" "WF_NORM_BOUND_PASS=1" "WF_GEMM_MFMA=16"; do
  env $sw python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -m gpu -q -x -k "not equals_small_tile" > gpurun_out/r3/p_switch_$sw.log 2>&1; echo "$sw rc=$?"; tail -2 gpurun_out/r3/p_switch_$sw.log
done
