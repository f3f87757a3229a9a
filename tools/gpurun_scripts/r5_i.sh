#!/bin/bash
# round 5, i: one rank of 8 under a BANDWIDTH MODEL of the interconnect (bench.py --emulate-comm: every collective = its local copies + a
# stream-ordered delay of latency + bytes / rate on its own stream) -- NOT a measurement: what each exchange mode would expose if RCCL's
# all-gather delivered 330 GB/s per rank (the figure DESIGN section 6 reasons with), half of that, or the all-pairs ideal; per-source
# broadcasts at one xGMI link (153 GB/s).  The exchange calibration SELECTS under the model.   -> gpurun_out/r5/i_*
mkdir -p gpurun_out/r5
for spec in "ag330:330,153,20" "ag165:165,153,30" "ag800:800,153,10"; do
  name=${spec%%:*}; m=${spec#*:}
  timeout 600 python bench.py --as-rank-of 8 --emulate-comm $m --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/i_wan_$name.json 2> gpurun_out/r5/i_wan_$name.err; echo "wan $name rc=$?"
done
timeout 600 python bench.py --as-rank-of 8 --emulate-comm 165,153,30 --exchange lockstep --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/i_wan_ag165_lockstep.json 2> gpurun_out/r5/i_wan_ag165_lockstep.err; echo "wan ag165 lockstep rc=$?"
timeout 600 python bench.py --workload longcat --distill --as-rank-of 8 --emulate-comm 330,153,20 --steps 4 --no-cpu-baseline > gpurun_out/r5/i_longcat_distill_ag330.json 2> gpurun_out/r5/i_longcat_distill_ag330.err; echo "longcat distill rc=$?"
timeout 600 python bench.py --workload longcat --as-rank-of 8 --emulate-comm 330,153,20 --steps 3 --no-cpu-baseline > gpurun_out/r5/i_longcat_cfg_ag330.json 2> gpurun_out/r5/i_longcat_cfg_ag330.err; echo "longcat cfg rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/i_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "g/p ms", round(d.get("guided_step_ms") or 0), round(d.get("plain_step_ms") or 0, 1), "model", d.get("comm_model", {}).get("allgather_gbps"))
        ex = d.get("exchange") or {}
        print("   selected", ex.get("selected"), "est", {k: round(v, 1) for k, v in (ex.get("estimated_ms_per_evaluation") or {}).items()})
        print("   per_rank", d.get("per_rank"), "exposed frac", d.get("comm_exposed_frac_of_layer"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r5/i_wan_ag330.err
