#!/bin/bash
# round 6, e: the whole GPU suite with the per-call tolerance keys (-> tests/golden/tolerances_mi355x.json), the driver's bench command, and
# the per-rank compute ceilings on the same box (one simulated rank of 8: lock-step pair, chunked2, 2 x 4 CFG groups; LongCat distilled)
#   -> gpurun_out/r6/e_*
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/r6/e_tolerances.txt
WF_TOL_LOG=$R/gpurun_out/r6/e_tolerances.txt timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r6/e_pytest.log; tail -5 gpurun_out/r6/e_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/e_bench.json 2> gpurun_out/r6/e_bench.err; echo "bench rc=$?"
for spec in "1:--no-also" "8:--as-rank-of 8" "8c2:--as-rank-of 8 --exchange chunked2" "2x4g:--as-rank-of 8 --exchange cfg2+gather"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $args --no-cpu-baseline > gpurun_out/r6/e_asrank_$name.json 2> gpurun_out/r6/e_asrank_$name.err; echo "asrank $name rc=$?"
done
timeout 600 python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r6/e_longcat_asrank8.json 2> gpurun_out/r6/e_longcat_asrank8.err; echo "longcat rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/e_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "norm", d.get("value_normalised"), "calib", (d.get("box_calib_tflops") or {}).get("mean"), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"), "roofline", (d.get("roofline") or {}).get("frac"), "hbm", (d.get("hbm") or {}).get("frac_of_8TBps"))
        for a in d.get("also", []):
            print("   also:", {k: v for k, v in a.items() if k not in ("workload", "vae_precision", "steps_per_s_basis", "refine_720p")})
    except Exception as e:
        print(f, "unreadable", e)
PY
