#!/bin/bash
# rocprofv3 kernel-trace stats of the Qwen-Image (BASELINE config 5) bench command.  usage: r3_profile_qwen.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --config qwen1024 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/trace.log 2>&1
cd $R
for f in $(find $O/trace -name "*kernel_stats.csv"); do cp $f $O/qwen_kernel_stats.csv; done
grep -h '"metric"' $O/trace.log | head -1 > $O/bench_line_under_trace.json
head -12 $O/qwen_kernel_stats.csv | cut -c1-160
rm -rf $O/trace
