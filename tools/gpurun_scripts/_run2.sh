set -x
python -m pytest tests/test_gpu_vae.py tests/test_gpu_inject_kernels.py -x -q > gpurun_out/r2_pytest_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_b.log
tail -5 gpurun_out/r2_pytest_b.log
timeout 1500 python tools/vae_precision_study.py --oracle --out gpurun_out/r2_vae_study.json > gpurun_out/r2_vae_study.log 2>&1; echo "rc=$?" >> gpurun_out/r2_vae_study.log
tail -60 gpurun_out/r2_vae_study.log
