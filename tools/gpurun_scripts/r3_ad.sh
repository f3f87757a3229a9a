#!/bin/bash
# round 3, ad: FFN-up weight stored with 14 080 rows (zero padding) so that the GEMM runs its 320-feature tile: DiT tests, then the bench
# window and the 8-rank simulated rank with WF_FFN_PAD=0 / default on one box
set -x
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py tests/test_gpu_multirank.py tests/test_gpu_timed_kernel_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed" > gpurun_out/r3/ad_tests.txt
cat gpurun_out/r3/ad_tests.txt
for pad in 0 1; do
  WF_FFN_PAD=$pad python bench.py --no-cpu-baseline > gpurun_out/r3/ad_bench_pad$pad.json 2> gpurun_out/r3/ad_bench_pad$pad.err
  WF_FFN_PAD=$pad python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r3/ad_asrank8_pad$pad.json 2> gpurun_out/r3/ad_asrank8_pad$pad.err
done
python - <<'PY'
import json
for pad in (0,1):
    b=json.load(open(f'gpurun_out/r3/ad_bench_pad{pad}.json')); r=json.load(open(f'gpurun_out/r3/ad_asrank8_pad{pad}.json'))
    print('pad',pad,'1 gpu',round(b['value'],5),b['guided_step_ms'],b['plain_step_ms'],'| rank of 8',round(r['value'],4),r['guided_step_ms'],r['plain_step_ms'],'x%.2f'%(r['value']/b['value']))
PY
