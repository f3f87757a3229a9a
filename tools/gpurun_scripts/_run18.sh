timeout 1500 python bench.py --workload longcat > gpurun_out/r2_longcat_bench.json 2> gpurun_out/r2_longcat_bench.err; echo rc=$?
timeout 1500 python bench.py --workload longcat --distill --no-cpu-baseline > gpurun_out/r2_longcat_distill_bench.json 2>/dev/null; echo rc=$?
python - <<'PY'
import json
for n in ("longcat_bench","longcat_distill_bench"):
    b=json.loads(open(f'gpurun_out/r2_{n}.json').read().strip().split(chr(10))[-1]); print(n, round(b['value'],4), round(b['guided_step_ms']), round(b['plain_step_ms']), b.get('roofline',{}).get('achieved'))
PY
python tools/longcat_bench.py 2>&1 | grep -v amdgpu | tail -4
