#!/bin/bash
# same-box timing of placement variants of the 4 x 64 attention iteration (tools/ablate/build_attn.py).  usage: r3_attn_variants.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
{
echo "== product"; PYTHONPATH=$PWD ATT_REPS=4 timeout 200 python tools/bench_attention.py 2>&1 | grep "svdq_attention\|max"
for lib in tools/ablate/libsvdq_amd_attn_*.so; do
  echo "== $lib"; SVDQ_LIB=$PWD/$lib PYTHONPATH=$PWD ATT_REPS=4 timeout 200 python tools/bench_attention.py 2>&1 | grep "geometry 2\|vs 1"
done
echo "== product again"; PYTHONPATH=$PWD ATT_REPS=4 timeout 200 python tools/bench_attention.py 2>&1 | grep "svdq_attention"
} > $O/attn_variants.txt 2>&1
cat $O/attn_variants.txt
