#!/bin/bash
# the pipeline as the first / last rank of 8 (zero halos, group edges) and a whole 50-step job as a middle rank: no exceptions, finite outputs
mkdir -p gpurun_out
for r in 0 7; do python bench.py --as-rank-of 8 --as-rank $r --no-cpu-baseline > gpurun_out/asrank8_r$r.json 2> gpurun_out/asrank8_r$r.err; echo "rank $r rc=$?"; cut -c1-60 gpurun_out/asrank8_r$r.json; python -c "
import json;d=json.loads(open('gpurun_out/asrank8_r$r.json').read().splitlines()[0]);print(round(d['value'],4), round(d['guided_step_ms']), round(d['plain_step_ms']))"; done
python bench.py --as-rank-of 8 --steps 50 --warmup 0 --no-cpu-baseline > gpurun_out/asrank8_job50.json 2> gpurun_out/asrank8_job50.err; echo "job50 rc=$?"
python -c "
import json;d=json.loads(open('gpurun_out/asrank8_job50.json').read().splitlines()[0]);print('job50 as rank 4 of 8:', round(d['value'],4), round(d['guided_step_ms']), round(d['plain_step_ms']))"
