#!/bin/bash
# round 4: the four launches of a block at the schnell 512^2 token count (M = 1536) and at M = 1024 / 768 under every workgroup geometry.   usage: <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; L=nunchaku_amd/csrc/libsvdq_amd.so
for M in 1536 1024; do
  for shape in "3072 3072 0" "12288 3072 0" "3072 12288 2" "3072 9216 3"; do
    set -- $shape
    echo "== M $M K $1 N $2 fuse $3"
    timeout 120 $P --lib $L --shape $M $1 $2 --fuse $3 --geoms 0,1,2,3 --warm 300 --iters 40 | python3 -c "
import sys, json
for ln in sys.stdin:
    try: d = json.loads(ln)
    except Exception: continue
    if 'us' in d: print('   geometry %d  %7.2f us  %7.1f TOP/s  sum %s' % (d['geometry'], d['us'], d['TOPS'], d['sum'][:8]))
"
  done
done 2>&1 | tee $O/small_m.txt
