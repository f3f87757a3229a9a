python -m pytest tests -m gpu -q > gpurun_out/r2_pytest_full_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_full_a.log
tail -12 gpurun_out/r2_pytest_full_a.log
timeout 900 python bench.py > gpurun_out/r2_bench_d.json 2> gpurun_out/r2_bench_d.err; echo "bench rc=$?"; cat gpurun_out/r2_bench_d.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
