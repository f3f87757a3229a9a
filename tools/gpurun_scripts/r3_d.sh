set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_warp.py tests/test_gpu_thirdparty_goldens.py tests/test_infer_entry.py tests/test_gpu_multirank.py -m gpu -q -x --durations=5 > gpurun_out/r3/d_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/d_tests.log
tail -15 gpurun_out/r3/d_tests.log
python tools/cross_attn_bench.py > gpurun_out/r3/d_cross_attn_bench.txt 2>&1; cat gpurun_out/r3/d_cross_attn_bench.txt
python bench.py --no-cpu-baseline > gpurun_out/r3/d_bench.json 2> gpurun_out/r3/d_bench.err; echo "rc=$?"
WF_CROSS_FUSED=0 python bench.py --no-cpu-baseline > gpurun_out/r3/d_bench_unfused.json 2> gpurun_out/r3/d_bench_unfused.err; echo "rc=$?"
python - <<'PY'
import json
for n in ("d_bench","d_bench_unfused"):
    d=json.load(open(f"gpurun_out/r3/{n}.json")); print(n, round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]), d["roofline"]["achieved"])
PY
