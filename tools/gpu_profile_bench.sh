#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then HBM counters of the GEMM
mkdir -p gpurun_out/prof_bench
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_bench/fetch -o g -- python $R/tools/prof_gemm.py 4096 3072 9216 > $R/gpurun_out/prof_bench/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_bench/write -o g -- python $R/tools/prof_gemm.py 4096 3072 9216 > $R/gpurun_out/prof_bench/write.log 2>&1
cd $R
grep -h '"metric"' gpurun_out/prof_bench/trace.log | head -1
find gpurun_out/prof_bench -name "*.csv" | head
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/prof_bench/trace/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, list(rows[0].keys()))
    for r in rows[:14]: print({k:(v[:70] if k=='Name' else v) for k,v in r.items()})
for d in ('fetch','write'):
    for f in glob.glob(f'gpurun_out/prof_bench/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']:
                agg[(r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(d,k,'avg per dispatch',sum(v)/len(v), 'n',len(v))
PY
# keep the big raw traces out of the merge (64 MiB cap): only stats + counters
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
