#!/bin/bash
# round 5, e: LongCat CFG groups (test + per-rank ceilings), the whole 50-step job measured, one more (value, box_calib_tflops) pair
#   -> gpurun_out/r5/e_*
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "cfg_groups or lockstep" 2>&1 | tail -4
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r5/e_bench10.json 2> gpurun_out/r5/e_bench10.err; echo "bench10 rc=$?"
timeout 900 python bench.py --steps 50 --warmup 0 --no-also --no-cpu-baseline > gpurun_out/r5/e_job50.json 2> gpurun_out/r5/e_job50.err; echo "job50 rc=$?"
for spec in "lockstep:" "2x4:--exchange cfg2+chunked2" "2x4g:--exchange cfg2+gather"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py --workload longcat --as-rank-of 8 --steps 3 --no-cpu-baseline $args > gpurun_out/r5/e_longcat_cfg_asrank8_$name.json 2> gpurun_out/r5/e_longcat_cfg_asrank8_$name.err; echo "longcat $name rc=$?"
done
timeout 600 python bench.py --workload longcat --steps 3 --no-cpu-baseline > gpurun_out/r5/e_longcat_cfg_1gpu.json 2> gpurun_out/r5/e_longcat_cfg_1gpu.err; echo "longcat 1gpu rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/e_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "norm", d.get("value_normalised"), "calib", (d.get("box_calib_tflops") or {}).get("mean"), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"), "attn", (d.get("roofline") or {}).get("avg_launch_ms"))
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:600])
    except Exception as e:
        print(f, "unreadable", e)
PY
