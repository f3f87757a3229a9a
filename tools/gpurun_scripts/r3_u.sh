#!/bin/bash
# round 3, u: k_conv_w4 with / without the out-of-line rare block, same box, interleaved
mkdir -p gpurun_out/r3
out=gpurun_out/r3/u_conv_ab.txt; : > $out
for round in 1 2; do for v in conv_noexpect conv_expect; do
  echo "== $v (round $round)" >> $out
  WF_LIB=worldforge_amd/_lib/lab/libwf_hip_$v.so python tools/conv_bench.py 2>/dev/null | grep "slice-major" >> $out
done; done
cat $out
