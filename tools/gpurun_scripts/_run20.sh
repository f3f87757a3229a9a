python -m pytest tests/test_gpu_warp.py -m gpu -q -x > gpurun_out/r2_pytest_m.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_m.log
tail -30 gpurun_out/r2_pytest_m.log
