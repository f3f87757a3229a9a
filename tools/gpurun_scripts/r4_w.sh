#!/bin/bash
# round 4, w: config 3 (720p) and LongCat-Video bench lines at HEAD (BASELINE configs 3 and 4; the builder-run numbers of the round)
set -x
mkdir -p gpurun_out/r4
python bench.py --no-cpu-baseline --height 720 --width 1280 --steps 4 > gpurun_out/r4/w_c3_bench.json 2> gpurun_out/r4/w_c3_bench.err; echo "rc=$?"
python bench.py --workload longcat --steps 5 > gpurun_out/r4/w_longcat_bench.json 2> gpurun_out/r4/w_longcat_bench.err; echo "rc=$?"
python bench.py --workload longcat --distill --steps 5 --no-cpu-baseline > gpurun_out/r4/w_longcat_distill_bench.json 2> gpurun_out/r4/w_longcat_distill_bench.err; echo "rc=$?"
python tools/longcat_bench.py --refine > gpurun_out/r4/w_longcat_refine.txt 2>&1; tail -5 gpurun_out/r4/w_longcat_refine.txt
python - <<'PY'
import json
for n in ("w_c3_bench","w_longcat_bench","w_longcat_distill_bench"):
    try:
        d=json.load(open(f"gpurun_out/r4/{n}.json")); print(n, round(d["value"],4), d.get("guided_step_ms"), d.get("plain_step_ms"), d.get("roofline",{}).get("achieved"), d.get("job50_steps_per_s"), d.get("job16_steps_per_s"))
    except Exception as e: print(n, "ERR", e)
PY
