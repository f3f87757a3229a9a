#!/bin/bash
# round 4, f: cycle-counted ablations of the K loop of k_gemm_pp (tools/gemm_pp_cycles.py ablate; build the lab libraries first with
# `python tools/gemm_pp_cycles.py build`) -> gpurun_out/r4/f_gemm_ablate.md
mkdir -p gpurun_out/r4
timeout 900 python tools/gemm_pp_cycles.py ablate > gpurun_out/r4/f_gemm_ablate.md 2> gpurun_out/r4/f_gemm_ablate.err
cat gpurun_out/r4/f_gemm_ablate.md; tail -3 gpurun_out/r4/f_gemm_ablate.err
