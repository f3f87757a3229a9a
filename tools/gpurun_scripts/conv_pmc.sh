#!/bin/bash
# rocprofv3 PMC passes of the LDS-resident-patch convolution k_conv_w4 on the 96-wide three-term layer (tools/conv_once.py C=96 X3=1 F16=1: the fp16-operand instantiation the fp16x3 VAE runs; F16=0 for the bf16 one).
#   gpurun --timeout 1200 -- 'bash tools/gpurun_scripts/conv_pmc.sh'     -> gpurun_out/conv_pmc/summary.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/conv_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; C=${C:-96} X3=${X3:-1} F16=${F16:-1} N=2 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- python3 $R/tools/conv_once.py > $OUT/$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc fetch FETCH_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python tools/pmc_summary.py $OUT/mfma $OUT/wait $OUT/fetch $OUT/tcc --kernel k_conv_w4 > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/conv_pmc/mfma/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv_w4' in r['Kernel_Name']: print('conv dur ms', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
PY
find $OUT -type f -size +2M -delete
