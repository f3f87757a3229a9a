#!/bin/bash
# round 5, b: whole GPU suite at the commit that scales the fp16 weight operands (tolerance log -> the new bars), VAE timings, and the
# per-rank compute ceilings on ONE box: rank of 8 in lock-step (with every exchange mode's compute cost from the calibration), the 2 x 4
# CFG-group split, and the LongCat distilled job as one rank of 8
#   -> gpurun_out/r5/b_*
mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/b_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r5/b_tolerances.txt timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r5/b_pytest.log
tail -12 gpurun_out/r5/b_pytest.log
timeout 300 python tools/vae_bench.py > gpurun_out/r5/b_vae_bench.txt 2>&1; tail -6 gpurun_out/r5/b_vae_bench.txt
for spec in "1:--no-also" "8:--as-rank-of 8" "2x4:--as-rank-of 8 --exchange cfg2+chunked2" "2x4g:--as-rank-of 8 --exchange cfg2+gather" "8c2:--as-rank-of 8 --exchange chunked2" "4:--as-rank-of 4" "2:--as-rank-of 2" "1x2:--as-rank-of 2 --exchange cfg2+gather"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $args --no-cpu-baseline > gpurun_out/r5/b_asrank_$name.json 2> gpurun_out/r5/b_asrank_$name.err; echo "asrank $name rc=$?"
done
timeout 600 python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r5/b_longcat_asrank8.json 2> gpurun_out/r5/b_longcat_asrank8.err; echo "longcat rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/b_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "norm", d.get("value_normalised"), "calib", (d.get("box_calib_tflops") or {}).get("mean"), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"))
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:900])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r5/b_asrank_2x4.err
