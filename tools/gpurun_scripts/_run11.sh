python -m pytest tests/test_gpu_dit.py -m gpu -q -x -k "attention or rmsnorm or heads or dit" > gpurun_out/r2_pytest_i.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_i.log
tail -6 gpurun_out/r2_pytest_i.log
python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_attn_bench_a.log
