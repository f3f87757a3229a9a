set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_vae.py -m gpu -q -x --durations=5 > gpurun_out/r3/b_multirank.log 2>&1; echo "rc=$?" >> gpurun_out/r3/b_multirank.log
tail -15 gpurun_out/r3/b_multirank.log
python bench.py --no-cpu-baseline > gpurun_out/r3/b_bench1.json 2> gpurun_out/r3/b_bench1.err; echo "rc=$?"
python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/b_asrank8.json 2> gpurun_out/r3/b_asrank8.err; echo "rc=$?"
WF_TRACE=gpurun_out/r3/b_asrank8_trace.json python bench.py --no-cpu-baseline --as-rank-of 8 --steps 4 > gpurun_out/r3/b_asrank8_t.json 2> gpurun_out/r3/b_asrank8_t.err; echo "rc=$?"
python bench.py --no-cpu-baseline --as-rank-of 4 > gpurun_out/r3/b_asrank4.json 2> gpurun_out/r3/b_asrank4.err; echo "rc=$?"
python bench.py --no-cpu-baseline --as-rank-of 2 > gpurun_out/r3/b_asrank2.json 2> gpurun_out/r3/b_asrank2.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("b_bench1","b_asrank2","b_asrank4","b_asrank8"):
    d=json.load(open(f"gpurun_out/r3/{n}.json")); print(n, round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]), d.get("job50_steps_per_s"))
PY
tail -3 gpurun_out/r3/*.err
