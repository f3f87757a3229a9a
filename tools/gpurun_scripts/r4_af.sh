#!/bin/bash
# round 4, af: which GEMM kernels the VAE runs with the batched P . V products, and for how long (rocprofv3 kernel table of tools/vae_bench.py) -> gpurun_out/r4/af_*.csv
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in unbatched batched; do
  if [ $v = batched ]; then unset WF_VAE_ATTN_UNBATCHED; else export WF_VAE_ATTN_UNBATCHED=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/af_prof_$v -- python3 $R/tools/vae_bench.py > $R/gpurun_out/r4/af_$v.log 2>&1
  find $R/gpurun_out/r4/af_prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/r4/af_kernel_stats_$v.csv
  rm -rf $R/gpurun_out/r4/af_prof_$v
  echo "== $v"; grep -E "k_gemm|k_split3|k_softmax|k_transpose" $R/gpurun_out/r4/af_kernel_stats_$v.csv | cut -c1-200
done
