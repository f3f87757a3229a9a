python -m pytest tests/test_gpu_vae.py -m gpu -q -x > gpurun_out/r2_pytest_j.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_j.log
tail -4 gpurun_out/r2_pytest_j.log
python tools/attn_bench.py 2>&1 | grep -v amdgpu
WF_ATTN_NOMAX=1 python tools/attn_bench.py 2>&1 | grep -v amdgpu
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_e.json 2> gpurun_out/r2_bench_e.err; python -c "
import json; b=json.loads(open('gpurun_out/r2_bench_e.json').read().strip().split(chr(10))[-1]); print(b['value'], b['guided_step_ms'], b['plain_step_ms'], b['roofline']['achieved'])"
