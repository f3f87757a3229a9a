set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_timed_kernel_parity.py -m gpu -q -x > gpurun_out/r3/m_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/m_tests.log; tail -4 gpurun_out/r3/m_tests.log
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3/m_bench_$i.json 2>/dev/null
WF_NORM_BOUND_PASS=1 python bench.py --no-cpu-baseline > gpurun_out/r3/m_bench_pass_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/m_asrank8_$i.json 2>/dev/null
WF_NORM_BOUND_PASS=1 python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/m_asrank8_pass_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r3/m_*.json")):
    d=json.load(open(n)); print(n.split('/')[-1], round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]), round(d["roofline"]["achieved"]))
PY
