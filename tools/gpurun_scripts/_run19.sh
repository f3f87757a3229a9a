set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; L=32760 N=2 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/r2_pmc4/$name -o pmc -- python3 $R/tools/attn_once.py > $R/gpurun_out/r2_pmc4_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<'PY'
import csv, collections
for n in ("mfma","fetch","write","wait","tcc"):
    kt=[r for r in csv.DictReader(open(f'gpurun_out/r2_pmc4/{n}/pmc_kernel_trace.csv')) if 'k_attn' in r['Kernel_Name']]
    did=kt[-1]['Dispatch_Id']; dur=(int(kt[-1]['End_Timestamp'])-int(kt[-1]['Start_Timestamp']))/1e6
    vals=collections.defaultdict(float)
    for r in csv.DictReader(open(f'gpurun_out/r2_pmc4/{n}/pmc_counter_collection.csv')):
        if r['Dispatch_Id']==did: vals[r['Counter_Name']]+=float(r['Counter_Value'])
    print(n, kt[-1]['Kernel_Name'][:60], round(dur,3), dict(vals))
PY
