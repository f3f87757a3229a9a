#!/bin/bash
# round 6, d: after batching the mid-block attention over frames and producing the halo-padded operands in place: VAE / multi-rank / e2e tests,
# VAE timings (all modes, needed-columns decode), the per-rank probe incl. the injection round trip, rocprofv3 of the round trip on rank 4 of 8
#   -> gpurun_out/r6/d_*
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/r6/d_tolerances.txt
WF_TOL_LOG=$R/gpurun_out/r6/d_tolerances.txt timeout 1500 python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py tests/test_gpu_longcat_sampler.py tests/test_gpu_sampler.py tests/test_gpu_dit.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r6/d_pytest.log; tail -6 gpurun_out/r6/d_pytest.log
timeout 400 python tools/vae_bench.py > gpurun_out/r6/d_vae_bench.txt 2>&1; tail -12 gpurun_out/r6/d_vae_bench.txt
timeout 600 python tools/vae_rank_probe.py > gpurun_out/r6/d_vae_rank_probe.txt 2>&1; tail -6 gpurun_out/r6/d_vae_rank_probe.txt
cd /tmp && export TMPDIR=/tmp
for P in 8 1; do
  P=$P RANK_SIM=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6/d_prof_rt_$P -- python3 $R/tools/vae_rank_once.py roundtrip > $R/gpurun_out/r6/d_once_rt_$P.txt 2>&1
  f=$(find $R/gpurun_out/r6/d_prof_rt_$P -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/kernel_stats_md.py $f "rocprofv3 --kernel-trace --stats: 3 x injection round trip (decode of the needed columns -> blend -> encode) at C2, P=$P (rank 4 of 8 simulated when P=8), fp16x3" > $R/gpurun_out/r6/d_kernels_rt_$P.md
  rm -rf $R/gpurun_out/r6/d_prof_rt_$P
  tail -2 $R/gpurun_out/r6/d_once_rt_$P.txt
done
cd $R
head -30 gpurun_out/r6/d_kernels_rt_8.md | cut -c1-160
