python -m pytest tests -m gpu -q -x > gpurun_out/r2_pytest_full_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_full_b.log
tail -5 gpurun_out/r2_pytest_full_b.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_f.json 2> gpurun_out/r2_bench_f.err; echo "bench rc=$?"
WF_ATTN_TRACK_MAX=1 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_f_track.json 2>/dev/null
WF_ATTN_PRESCALE=0 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_f_noprescale.json 2>/dev/null
python - <<'PY'
import json
for n in ("f","f_track","f_noprescale"):
    b=json.loads(open(f'gpurun_out/r2_bench_{n}.json').read().strip().split(chr(10))[-1]); print(n, round(b['value'],4), round(b['guided_step_ms']), round(b['plain_step_ms']), round(b['roofline']['achieved']), b['roofline']['kernel'][:30])
PY
