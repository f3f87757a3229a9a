#!/bin/bash
# round 6, b: whole GPU suite (tolerance log), the driver's bench command with the new `also` windows (C2 with the TF32-class VAE, 720p
# with a warm-up step, LongCat on the bf16 VAE module), and the supervised 8-rank launch on one GPU over gloo (debug transport)
#   -> gpurun_out/r6/b_*
mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/b_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r6/b_tolerances.txt timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r6/b_pytest.log
tail -12 gpurun_out/r6/b_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/b_bench.json 2> gpurun_out/r6/b_bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r6/b_bench.err
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --layers 2 --no-cpu-baseline > gpurun_out/r6/b_bench8_gloo.json 2> gpurun_out/r6/b_bench8_gloo.err; echo "bench8 rc=$?"
grep "^\[bench" gpurun_out/r6/b_bench8_gloo.err | tail -20
python - <<'PY'
import json
for f in ("gpurun_out/r6/b_bench.json", "gpurun_out/r6/b_bench8_gloo.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, "value", d.get("value"), "g/p", d.get("guided_step_ms"), d.get("plain_step_ms"), "roofline", (d.get("roofline") or {}).get("frac"), "also_s", d.get("also_s"))
        for a in d.get("also", []):
            print("   also:", {k: v for k, v in a.items() if k not in ("workload", "vae_precision", "steps_per_s_basis")})
        for k in ("rccl", "collectives_used", "process_groups", "fallback"):
            if k in d: print("  ", k, d[k])
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:600])
    except Exception as e:
        print(f, "unreadable", e)
PY
