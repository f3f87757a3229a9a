#!/bin/bash
# round 5, g: EIGHT rank processes of bench.py through real torch.distributed collectives on one GPU (gloo, debug transport): the exchange
# calibration over all nine candidates incl. the 2 x 4 CFG groups, then forced cfg2+bcast (sub-group broadcasts: global-rank mapping) and
# LongCat with CFG (CFG groups first); small shapes, timings meaningless   -> gpurun_out/r5/g_*
mkdir -p gpurun_out/r5
export WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo
S="--layers 2 --frames 17 --height 240 --width 416 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 1200 python bench.py --gpus 8 $S > gpurun_out/r5/g_gloo8.json 2> gpurun_out/r5/g_gloo8.err; echo "gloo8 rc=$?"
timeout 1200 python bench.py --gpus 8 $S --exchange cfg2+bcast > gpurun_out/r5/g_gloo8_cfg2_bcast.json 2> gpurun_out/r5/g_gloo8_cfg2_bcast.err; echo "gloo8 cfg2+bcast rc=$?"
timeout 1200 python bench.py --gpus 4 --workload longcat $S > gpurun_out/r5/g_gloo4_longcat_cfg.json 2> gpurun_out/r5/g_gloo4_longcat_cfg.err; echo "gloo4 longcat rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/g_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "n_gpus", d.get("n_gpus"), d["config"]["parallelism"][:100])
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:900])
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -v Gloo gpurun_out/r5/g_gloo8.err | tail -3
