#!/bin/bash
# SQ / cache counters of the quantiser kernel at the FLUX shape (separate PMC passes, kernel-trace only)
O=gpurun_out/prof_quant; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
export PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $R/$O/sq1 -o a --output-format csv -- python $R/tools/bench_quant.py > $R/$O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_REQ TCP_REQ_MISS TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/$O/c1 -o a --output-format csv -- python $R/tools/bench_quant.py > $R/$O/c1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE TA_BUSY_avr -d $R/$O/sq2 -o a --output-format csv -- python $R/tools/bench_quant.py > $R/$O/sq2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('sq1','c1','sq2'):
    for f in glob.glob(f'gpurun_out/prof_quant/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'quantize_kernel' in r['Kernel_Name'] and r.get('Grid_Size', '') :
                agg[(r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d,k,sum(v)/len(v), len(v))
PY
tail -8 $O/sq1.log
find $O -name "*.csv" -size +2M -delete
