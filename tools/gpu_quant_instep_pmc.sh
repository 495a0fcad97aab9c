#!/bin/bash
# the quantiser INSIDE the denoise step: SQ / cache counters per launch (separate PMC passes), to compare with tools/gpu_quant_pmc.sh
O=gpurun_out/prof_quant_step; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/$O/sq1 -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $R/$O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/$O/c1 -o a --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $R/$O/c1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ('sq1','c1'):
    for f in glob.glob(f'gpurun_out/prof_quant_step/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'quantize_kernel' in r['Kernel_Name']:
                agg[(r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(d,k,round(sum(v)/len(v),1), len(v))
    for f in glob.glob(f'gpurun_out/prof_quant_step/{d}/**/*kernel_trace.csv', recursive=True):
        dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'quantize_kernel' in r['Kernel_Name']]
        if dur: print(d,'quantize kernel durations: n',len(dur),'avg %.2f us min %.2f max %.2f'%(sum(dur)/len(dur),min(dur),max(dur)))
PY
find $O -name "*.csv" -size +2M -delete
