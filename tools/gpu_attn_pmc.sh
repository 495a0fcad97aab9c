#!/bin/bash
# SQ / LDS counters of the attention kernel (separate PMC passes, kernel-trace only)
mkdir -p gpurun_out/prof_attn
export TMPDIR=/tmp
R=$PWD
export PYTHONPATH=$R
V="${1:-2048}"
cd /tmp
SVDQ_ATT_DEBUG=$V rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $R/gpurun_out/prof_attn/sq1 -o a --output-format csv -- python $R/tools/bench_attention.py > $R/gpurun_out/prof_attn/sq1.log 2>&1
SVDQ_ATT_DEBUG=$V rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d $R/gpurun_out/prof_attn/sq2 -o a --output-format csv -- python $R/tools/bench_attention.py > $R/gpurun_out/prof_attn/sq2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('sq1','sq2'):
    for f in glob.glob(f'gpurun_out/prof_attn/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'attention' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d,k,sum(v)/len(v), len(v))
PY
tail -3 gpurun_out/prof_attn/sq2.log
find gpurun_out/prof_attn -name "*.csv" -size +2M -delete
