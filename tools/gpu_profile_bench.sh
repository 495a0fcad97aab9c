#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then the HBM-side counters of
# every gemm_w4a4 dispatch of the same command (separate --pmc passes, MI355X_MICROARCH.md "HBM")
mkdir -p gpurun_out/prof_bench
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_bench/fetch -o g -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_bench/write -o g -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/write.log 2>&1
cd $R
grep -h '"metric"' gpurun_out/prof_bench/trace.log | head -1
python - <<'PY'
import csv, glob, collections, json
for f in glob.glob('gpurun_out/prof_bench/trace/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print({k:(v[:60] if k=='Name' else v) for k,v in r.items() if k in ('Name','Calls','AverageNs','Percentage')})
out={}
for d in ('fetch','write'):
    for f in glob.glob(f'gpurun_out/prof_bench/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']:
                agg[(r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in agg.items():
            print(d,k,'avg per gemm dispatch',sum(v)/len(v), 'n',len(v)); out[k]={'avg_per_dispatch_KB':sum(v)/len(v),'dispatches':len(v)}
json.dump(out, open('gpurun_out/prof_bench/gemm_hbm_counters.json','w'), indent=1)
PY
# keep the big raw traces out of the merge (64 MiB cap): only stats + summaries
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
find gpurun_out/prof_bench -name "*counter_collection.csv" -size +8M -delete
