#!/bin/bash
# round 5, c: the new kernel-level exchange tests + the VAE tests at the tightened bars, LongCat as one rank of 8 (distilled: every exchange
# mode's compute cost; CFG: the lock-step pair against sequential samples), the two-depth exchange calibration of the Wan job as one rank
# of 8, and the PMC passes of the timed attention kernel at HEAD (its set-up code changed: seg_stride / second window)
#   -> gpurun_out/r5/c_*, gpurun_out/attn_pmc/
mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/c_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r5/c_tolerances.txt timeout 1200 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_longcat.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r5/c_pytest.log
tail -6 gpurun_out/r5/c_pytest.log
timeout 600 python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r5/c_longcat_distill_asrank8.json 2> gpurun_out/r5/c_longcat_distill_asrank8.err; echo "longcat distill rc=$?"
timeout 600 python bench.py --workload longcat --as-rank-of 8 --steps 3 --no-cpu-baseline > gpurun_out/r5/c_longcat_cfg_asrank8.json 2> gpurun_out/r5/c_longcat_cfg_asrank8.err; echo "longcat cfg rc=$?"
timeout 600 python bench.py --workload longcat --as-rank-of 8 --steps 3 --no-cpu-baseline --exchange gather > gpurun_out/r5/c_longcat_cfg_asrank8_sequential.json 2> gpurun_out/r5/c_longcat_cfg_asrank8_sequential.err; echo "longcat cfg sequential rc=$?"
timeout 600 python bench.py --as-rank-of 8 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/c_asrank8_calib2.json 2> gpurun_out/r5/c_asrank8_calib2.err; echo "asrank8 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/c_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"))
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:1500])
    except Exception as e:
        print(f, "unreadable", e)
PY
bash tools/gpurun_scripts/attn_pmc.sh 2>&1 | tail -25
