set -x
export WF_LIB=$GRAFT_REPO_ROOT/worldforge_amd/_lib/libwf_hip_convtiming.so
for l in 0 1; do for x in 0 1; do LAYOUT=$l X3=$x python tools/conv_timing.py; done; done > gpurun_out/r2_conv_timing_a.log 2>&1
cat gpurun_out/r2_conv_timing_a.log
unset WF_LIB
python -m pytest tests/test_gpu_vae.py tests/test_gpu_e2e.py tests/test_trace.py tests/test_config1_truck.py tests/test_gpu_multirank.py tests/test_gpu_config3.py tests/test_gpu_dit.py -m gpu -q -x > gpurun_out/r2_pytest_e.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_e.log
tail -8 gpurun_out/r2_pytest_e.log
