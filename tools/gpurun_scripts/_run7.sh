set -x
R=$GRAFT_REPO_ROOT
for x in 0 1; do WF_LIB=$R/worldforge_amd/_lib/libwf_hip_convtiming.so LAYOUT=1 X3=$x python tools/conv_timing.py; done > gpurun_out/r2_conv_timing_b.log 2>&1
grep -v amdgpu gpurun_out/r2_conv_timing_b.log
timeout 900 python bench.py > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err; echo "bench rc=$?"; cat gpurun_out/r2_bench_c.json
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; C=96 X3=1 N=2 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/r2_pmc_conv/$name -o pmc -- python3 $R/tools/conv_once.py > $R/gpurun_out/r2_pmc_conv_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc fetch FETCH_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python tools/pmc_summary.py gpurun_out/r2_pmc_conv/* --kernel k_conv_w4 > gpurun_out/r2_pmc_conv_summary.txt 2>&1; cat gpurun_out/r2_pmc_conv_summary.txt
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r2_pmc_conv/mfma/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_conv_w4' in r['Kernel_Name']: print('conv dur ms',(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
