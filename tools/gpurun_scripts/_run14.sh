python -m pytest tests/test_gpu_dit.py -m gpu -q -x -k "attention or rmsnorm or heads" > gpurun_out/r2_pytest_k.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_k.log
tail -6 gpurun_out/r2_pytest_k.log
python tools/attn_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2_attn_bench_b.log
