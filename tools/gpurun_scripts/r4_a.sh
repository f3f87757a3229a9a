#!/bin/bash
# round 4, a: issue-slot lab (tools/gemm_lab/issue_lab.hip): free issue slots beside 32x32x16 vs 16x16x32 MFMA streams -> gpurun_out/r4/a_issue_lab.md
mkdir -p gpurun_out/r4
timeout 300 tools/gemm_lab/issue_lab > gpurun_out/r4/a_issue_lab.md 2> gpurun_out/r4/a_issue_lab.err
cat gpurun_out/r4/a_issue_lab.md
