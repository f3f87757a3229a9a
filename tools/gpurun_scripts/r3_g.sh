set -x
mkdir -p gpurun_out/r3
which rocm-smi; rocm-smi --showpower --showclocks --json 2>&1 | head -c 600
python tools/conv_bench.py > gpurun_out/r3/g_conv_bench.txt 2>&1; cat gpurun_out/r3/g_conv_bench.txt
python tools/gemm_energy.py > gpurun_out/r3/g_gemm_energy.md 2> gpurun_out/r3/g_gemm_energy.err; cat gpurun_out/r3/g_gemm_energy.md; tail -3 gpurun_out/r3/g_gemm_energy.err
bash tools/gpurun_scripts/attn_pmc.sh > gpurun_out/r3/g_attn_pmc.log 2>&1; tail -30 gpurun_out/r3/g_attn_pmc.log
