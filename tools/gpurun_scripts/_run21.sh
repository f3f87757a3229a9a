python -m pytest tests/test_gpu_rccl2.py -m gpu -q -x > gpurun_out/r2_pytest_n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_n.log
tail -30 gpurun_out/r2_pytest_n.log
