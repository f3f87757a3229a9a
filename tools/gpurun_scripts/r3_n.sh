set -x
mkdir -p gpurun_out/r3
( time python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r3/n_gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r3/n_gpu_suite.log
tail -16 gpurun_out/r3/n_gpu_suite.log
