#!/bin/bash
# round 5, d: the WHOLE N > 1 path of bench.py through real torch.distributed collectives on a one-GPU box (ranks share GPU 0 over gloo: a
# debug transport, timings meaningless): self-launch, Comm.split, the exchange calibration over every candidate (lock-step pair, chunked,
# bcast, gather, CFG groups), the selected mode through the timed window, per-rank figures -- 2 and 4 ranks, Wan and LongCat, small shapes
#   -> gpurun_out/r5/d_*
mkdir -p gpurun_out/r5
export WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo
S="--layers 4 --frames 17 --height 240 --width 416 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 900 python bench.py --gpus 2 $S > gpurun_out/r5/d_gloo2.json 2> gpurun_out/r5/d_gloo2.err; echo "gloo2 rc=$?"
timeout 900 python bench.py --gpus 4 $S > gpurun_out/r5/d_gloo4.json 2> gpurun_out/r5/d_gloo4.err; echo "gloo4 rc=$?"
timeout 900 python bench.py --gpus 4 $S --exchange cfg2+chunked2 > gpurun_out/r5/d_gloo4_cfg2.json 2> gpurun_out/r5/d_gloo4_cfg2.err; echo "gloo4 cfg2 rc=$?"
timeout 900 python bench.py --gpus 4 $S --exchange bcast > gpurun_out/r5/d_gloo4_bcast.json 2> gpurun_out/r5/d_gloo4_bcast.err; echo "gloo4 bcast rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --workload longcat --layers 4 --frames 17 --height 240 --width 416 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/d_torchrun2_longcat.json 2> gpurun_out/r5/d_torchrun2_longcat.err; echo "torchrun2 longcat rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29562 bench.py --gpus 2 --workload longcat --distill --layers 4 --frames 17 --height 240 --width 416 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/d_torchrun2_longcat_distill.json 2> gpurun_out/r5/d_torchrun2_longcat_distill.err; echo "torchrun2 longcat distill rc=$?"
unset WF_SHARE_GPU WF_COMM_BACKEND
timeout 600 python -m pytest tests/test_gpu_rccl2.py tests/test_gpu_vae.py::test_f16_range_flag_is_raised_and_reported tests/test_gpu_vae.py::test_vae_decode_returns_while_the_gpu_is_still_busy -m gpu -q 2>&1 | tail -4
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/d_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "n_gpus", d.get("n_gpus"), d["config"]["parallelism"][:90])
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"])[:700])
        if d.get("per_rank"): print("   per_rank", json.dumps(d["per_rank"])[:500])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -4 gpurun_out/r5/d_gloo4.err
