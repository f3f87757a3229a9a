python -m pytest tests/test_gpu_bsa.py tests/test_gpu_longcat.py tests/test_gpu_dit.py -m gpu -q -x > gpurun_out/r2_pytest_l.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_l.log
tail -4 gpurun_out/r2_pytest_l.log
python tools/longcat_bench.py --refine > gpurun_out/r2_longcat_refine.log 2>&1; tail -6 gpurun_out/r2_longcat_refine.log
WF_GEMM_KERNEL=w4 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_g_w4gemm.json 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_g.json 2>/dev/null
python - <<'PY'
import json
for n in ("g","g_w4gemm"):
    b=json.loads(open(f'gpurun_out/r2_bench_{n}.json').read().strip().split(chr(10))[-1]); print(n, round(b['value'],4), round(b['guided_step_ms']), round(b['plain_step_ms']), round(b['roofline']['achieved']))
PY
