#!/bin/bash
# bench line + per-kernel time breakdown of the same command (kernel-trace stats only)
mkdir -p gpurun_out/prof_bench
export TMPDIR=/tmp
R=$PWD
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench/trace.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_bench/trace/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:22]: print(r['Name'][:70], r['Calls'], r['AverageNs'][:9], r['Percentage'])
PY
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
