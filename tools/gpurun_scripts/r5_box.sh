#!/bin/bash
# round 5: one (value, box_calib_tflops) pair from whatever box this call lands on (do lines from boxes >= 4 % apart agree
# on value_normalised?)   usage: gpurun -- 'bash tools/gpurun_scripts/r5_box.sh TAG'   -> gpurun_out/r5/box_TAG.json
mkdir -p gpurun_out/r5
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r5/box_$1.json 2> gpurun_out/r5/box_$1.err; echo "rc=$?"
python - <<PY
import json
d = json.loads(open("gpurun_out/r5/box_$1.json").read().strip().splitlines()[-1])
print("value", d["value"], "norm", d.get("value_normalised"), "calib", d.get("box_calib_tflops"), "attn ms", d["roofline"]["avg_launch_ms"])
PY
