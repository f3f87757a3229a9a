#!/bin/bash
# round 6, c: where the row-sharded VAE of one simulated rank of 8 spends its time -- vae_rank_probe (per entry point) and rocprofv3 kernel
# traces of decode and encode on rank 4 of 8, next to the single-GPU traces; the whole GPU suite with the per-case tolerance log; decode of the needed columns
#   -> gpurun_out/r6/c_*
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/r6/c_tolerances.txt
WF_TOL_LOG=$R/gpurun_out/r6/c_tolerances.txt timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r6/c_pytest.log; tail -6 gpurun_out/r6/c_pytest.log
timeout 400 python tools/vae_bench.py fp32 fp16 > gpurun_out/r6/c_vae_bench.txt 2>&1; tail -8 gpurun_out/r6/c_vae_bench.txt
timeout 400 python tools/vae_rank_probe.py > gpurun_out/r6/c_vae_rank_probe.txt 2>&1; tail -4 gpurun_out/r6/c_vae_rank_probe.txt
cd /tmp && export TMPDIR=/tmp
for what in decode encode; do
  for P in 8 1; do
    P=$P RANK_SIM=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6/c_prof_${what}_$P -- python3 $R/tools/vae_rank_once.py $what > $R/gpurun_out/r6/c_once_${what}_$P.txt 2>&1
    f=$(find $R/gpurun_out/r6/c_prof_${what}_$P -name "*kernel_stats.csv" | head -1)
    python3 $R/tools/kernel_stats_md.py $f "rocprofv3 --kernel-trace --stats: 3 x VAE $what at C2, P=$P (rank 4 of 8 simulated when P=8), fp16x3" > $R/gpurun_out/r6/c_kernels_${what}_$P.md
    rm -rf $R/gpurun_out/r6/c_prof_${what}_$P
    tail -2 $R/gpurun_out/r6/c_once_${what}_$P.txt
  done
done
cd $R
head -40 gpurun_out/r6/c_kernels_decode_8.md
