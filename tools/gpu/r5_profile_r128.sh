#!/bin/bash
# round 5: rocprofv3 kernel-trace stats of the rank-128 lines (which kernels a rank-128 step runs: the all-rank GEMM kernels, the pack kernels, the solo-carry
# GELU_QUANT kernel, the quantiser's multi-slab fast path).   usage: r5_profile_r128.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_dev -o bench -- python $R/bench.py --rank 128 --steps 3 --warmup 1 --prof-steps 1 --no-cpu-baseline > $R/$O/trace_dev.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_qwen -o bench -- python $R/bench.py --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128 --steps 3 --warmup 1 --prof-steps 1 --no-cpu-baseline > $R/$O/trace_qwen.log 2>&1
cd $R
for t in dev qwen; do f=$(find $O/trace_$t -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats_r128_$t.csv; head -12 $O/bench_kernel_stats_r128_$t.csv | cut -c1-150; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace_dev $O/trace_qwen
