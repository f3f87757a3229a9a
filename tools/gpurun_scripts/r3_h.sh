set -x
mkdir -p gpurun_out/r3
( time python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r3/h_gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r3/h_gpu_suite.log
tail -30 gpurun_out/r3/h_gpu_suite.log
python tools/gemm_energy.py > gpurun_out/r3/h_gemm_energy.md 2> gpurun_out/r3/h_gemm_energy.err; cat gpurun_out/r3/h_gemm_energy.md
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/h_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3/h_smoke.log
