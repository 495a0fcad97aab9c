#!/bin/bash
# end of round 3, last call: all GPU tests, smoke(), kernel-trace stats of the bench command, the default bench line.  usage: r3_final2.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/trace.log 2>&1)
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -6 $O/bench_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
