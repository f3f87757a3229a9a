set -x
python tools/conv_bench.py > gpurun_out/r2_conv_bench_b.log 2>&1; grep -v amdgpu gpurun_out/r2_conv_bench_b.log
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py -m gpu -q -x -k "conv or vae or fp32" > gpurun_out/r2_pytest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_f.log
tail -4 gpurun_out/r2_pytest_f.log
python tools/vae_precision_study.py --fixture tests/golden/g17_schedule_length_oracle.npz --out gpurun_out/r2_vae_study_v2.json > gpurun_out/r2_vae_study_v2.log 2>&1; echo "study rc=$?"
tail -c 6000 gpurun_out/r2_vae_study_v2.log
python -m pytest tests/test_gpu_schedule_length.py -m gpu -q -x -s > gpurun_out/r2_pytest_g.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_g.log
tail -12 gpurun_out/r2_pytest_g.log
