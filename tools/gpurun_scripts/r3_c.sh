set -x
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
( time python -m pytest tests/test_gpu_schedule_length.py -m gpu -q -s --durations=6 ) > gpurun_out/r3/c_schedlen.log 2>&1; echo "rc=$?" >> gpurun_out/r3/c_schedlen.log
grep -v "^$" gpurun_out/r3/c_schedlen.log | tail -40
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo timeout 600 python tools/comm_probe.py --gpus 2 --iters 3 --layers 2 > gpurun_out/r3/c_comm_probe_gloo.json 2> gpurun_out/r3/c_comm_probe_gloo.err; echo "probe rc=$?"
cat gpurun_out/r3/c_comm_probe_gloo.json; tail -5 gpurun_out/r3/c_comm_probe_gloo.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/c_prof8 -- python3 $R/bench.py --no-cpu-baseline --as-rank-of 8 --steps 4 > $R/gpurun_out/r3/c_asrank8_prof.json 2> $R/gpurun_out/r3/c_asrank8_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/c_prof1 -- python3 $R/bench.py --no-cpu-baseline --steps 4 > $R/gpurun_out/r3/c_bench1_prof.json 2> $R/gpurun_out/r3/c_bench1_prof.err
cd $R
for d in c_prof8 c_prof1; do find gpurun_out/r3/$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3/${d}_kernel_stats.csv; rm -rf gpurun_out/r3/$d; done
head -30 gpurun_out/r3/c_prof8_kernel_stats.csv
