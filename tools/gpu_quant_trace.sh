#!/bin/bash
O=gpurun_out/quant_trace; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
export PYTHONPATH=$R
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O -o q --output-format csv -- python $R/tools/bench_quant.py > $R/$O/log.txt 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/quant_trace/**/*kernel_trace.csv', recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'quantize' in n or 'fill' in n.lower() or 'memset' in n.lower():
            agg[(n[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in sorted(agg.items()): print(k, f"n={len(v)} avg={sum(v)/len(v):.2f} us min={min(v):.2f}")
PY
tail -7 $O/log.txt
find $O -name "*.csv" -size +2M -delete
