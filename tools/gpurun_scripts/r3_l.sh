set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=5 > gpurun_out/r3/l_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/l_tests.log; tail -12 gpurun_out/r3/l_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r3/l_bench.json 2>/dev/null
WF_NORM_BOUND_PASS=1 python bench.py --no-cpu-baseline > gpurun_out/r3/l_bench_pass.json 2>/dev/null
python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/l_asrank8.json 2>/dev/null
WF_NORM_BOUND_PASS=1 python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/l_asrank8_pass.json 2>/dev/null
python - <<'PY'
import json
for n in ("l_bench","l_bench_pass","l_asrank8","l_asrank8_pass"):
    d=json.load(open(f"gpurun_out/r3/{n}.json")); print(n, round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]), round(d["roofline"]["achieved"]))
PY
