#!/bin/bash
# round 5: the split kernel with its weight fragments double-buffered: the tests that cover it + launch-level times (probe) + per-kernel times of one rank-128 step.  usage: r5_split_ab3.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_parity.py -q -k "fused_output_quantiser or next_low_rank_down" > $O/pytest_small.txt 2>&1; tail -3 $O/pytest_small.txt
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
for c in "4608 128 128 0 6,7" "6400 128 128 256 7" "4608 48 48 0 1,7" "4608 160 160 0 7"; do
  set -- $c
  echo "fc1 M=$1 R=$2 R2=$3 split_rows=$4"
  timeout 120 $P --lib $PL --shape $1 3072 12288 --fuse 2 --R $2 --R2 $3 --split $4 --geoms $5 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: print(l.strip()[:160]); continue
    if 'us' in r: print('  geo=%s %.1f us %.3f GHz' % (r.get('geometry'), r['us'], r.get('eff_GHz',0)))"
done
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_dev -o bench -- python $R/bench.py --rank 128 --steps 3 --warmup 1 --prof-steps 1 --no-cpu-baseline > $R/$O/trace_dev.log 2>&1
cd $R
f=$(find $O/trace_dev -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats_r128_dev.csv; grep "lowrank_down_split\|pack_lora\|attention_kernel\|gemm_w4a4_kernel<0, 2" $O/bench_kernel_stats_r128_dev.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace_dev
