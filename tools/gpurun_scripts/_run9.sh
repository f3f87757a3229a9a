python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py -m gpu -q -x -k "conv or vae or fp32" > gpurun_out/r2_pytest_h.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_h.log
tail -4 gpurun_out/r2_pytest_h.log
python tools/conv_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_conv_bench_c.log
