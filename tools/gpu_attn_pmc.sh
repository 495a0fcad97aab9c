#!/bin/bash
# SQ / LDS counters of the attention kernel at the FLUX.1 shape (separate PMC passes, kernel-trace only); tools/bench_attention.py
# runs the plain grid and the persistent schedule alternately: both kernels appear
O=gpurun_out/prof_attn; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
export PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $R/$O/sq1 -o a --output-format csv -- python $R/tools/bench_attention.py > $R/$O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/$O/sq2 -o a --output-format csv -- python $R/tools/bench_attention.py > $R/$O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/$O/c1 -o a --output-format csv -- python $R/tools/bench_attention.py > $R/$O/c1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('sq1','sq2','c1'):
    for f in glob.glob(f'gpurun_out/prof_attn/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'attention_kernel' in r['Kernel_Name']:
                kind = 'persistent' if 'true>' in r['Kernel_Name'] else 'plain'
                agg[(kind, r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d,k[0],k[1],round(sum(v)/len(v),1), len(v))
PY
find $O -name "*.csv" -size +2M -delete
