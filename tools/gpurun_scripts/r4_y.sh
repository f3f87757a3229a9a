#!/bin/bash
# round 4, y: per-rank compute ceilings at HEAD: the bench window as one rank of 2 / 4 / 8 (collectives served locally) and on one GPU, same box
set -x
mkdir -p gpurun_out/r4
python bench.py --no-cpu-baseline > gpurun_out/r4/y_bench_1gpu.json 2> gpurun_out/r4/y_bench_1gpu.err
for n in 2 4 8; do python bench.py --no-cpu-baseline --as-rank-of $n > gpurun_out/r4/y_asrank_$n.json 2> gpurun_out/r4/y_asrank_$n.err; done
python - <<'PY'
import json
b=json.load(open('gpurun_out/r4/y_bench_1gpu.json'))
print('1 gpu', b['value'], b['guided_step_ms'], b['plain_step_ms'])
for n in (2,4,8):
    d=json.load(open(f'gpurun_out/r4/y_asrank_{n}.json')); print(n, d['value'], d['guided_step_ms'], d['plain_step_ms'], 'x%.2f'%(d['value']/b['value']))
PY
