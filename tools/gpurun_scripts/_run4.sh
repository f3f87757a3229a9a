set -x
R=$GRAFT_REPO_ROOT
python tools/conv_bench.py > gpurun_out/r2_conv_bench_a.log 2>&1; cat gpurun_out/r2_conv_bench_a.log
python -m pytest tests/test_gpu_vae.py tests/test_gpu_e2e.py tests/test_gpu_trace.py tests/test_trace.py tests/test_config1_truck.py tests/test_gpu_multirank.py -m gpu -q -x > gpurun_out/r2_pytest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_d.log
tail -5 gpurun_out/r2_pytest_d.log
cd /tmp && export TMPDIR=/tmp
pmc() { # name, counters...
  name=$1; shift
  L=32760 N=2 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/r2_pmc/$name -o pmc -- python3 $R/tools/attn_once.py > $R/gpurun_out/r2_pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python tools/pmc_summary.py gpurun_out/r2_pmc/* --kernel k_attn > gpurun_out/r2_pmc_summary.txt 2>&1; cat gpurun_out/r2_pmc_summary.txt
find gpurun_out/r2_pmc -name "*.csv" | head -20
