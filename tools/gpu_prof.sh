#!/bin/bash
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o gemm -- python $R/tools/prof_gemm.py > $R/gpurun_out/prof/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA -d $R/gpurun_out/prof/pmc1 -o gemm -- python $R/tools/prof_gemm.py > $R/gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM -d $R/gpurun_out/prof/pmc2 -o gemm -- python $R/tools/prof_gemm.py > $R/gpurun_out/prof/pmc2.log 2>&1
cd $R
find gpurun_out/prof -name "*.csv" | head -20
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/prof/trace/**/*kernel_stats.csv', recursive=True):
    print(f); print(open(f).read()[:1500])
for d in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/prof/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(d,k,sum(v)/len(v), len(v))
PY
