#!/bin/bash
# rocprofv3 PMC passes of the timed self-attention kernel k_attn_w4<4> at the C2 shape (one pass per counter group, program directly after `--`,
# never mixed with trace domains) -> gpurun_out/attn_pmc/, profiles-ready summary + profiles/attn_pmc_latest.json (the roofline side fields of bench.py).
#   gpurun --timeout 1200 -- 'bash tools/gpurun_scripts/attn_pmc.sh'     then copy gpurun_out/attn_pmc/{summary.txt,attn_pmc_latest.json} to profiles/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/attn_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; L=32760 N=2 MODE=${MODE:-ps} timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- python3 $R/tools/attn_once.py > $OUT/$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
DUR=$(python - <<'PY'
import csv, glob
d = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for f in glob.glob('gpurun_out/attn_pmc/mfma/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f)) if 'k_attn_w4' in r['Kernel_Name'] ]
print(d[-1] if d else '')
PY
)
PMC_DURATION_MS=$DUR python tools/pmc_summary.py $OUT/mfma $OUT/fetch $OUT/write $OUT/wait $OUT/tcc --kernel "k_attn_w4<4>" --attn-json $OUT/attn_pmc_latest.json \
  --tokens 32760 --heads 40 --source "gpurun_out/attn_pmc (tools/gpurun_scripts/attn_pmc.sh), warm launch, ${DUR} ms under the profiler" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -type f -size +2M -delete
