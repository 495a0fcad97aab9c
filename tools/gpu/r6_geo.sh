#!/bin/bash
# round 6: geometry choice re-checked after the loop changes (isolated launches, M = 4608, rank 32): QKV + RoPE and fc1 + GELU_QUANT under geometry 0 (the library's), 1, 2
for rep in 1 2; do
for s in "4608 3072 9216 3" "4608 3072 12288 2"; do set -- $s
  timeout 200 tools/ablate/gemm_probe --shape $1 $2 $3 --fuse $4 --R 32 --R2 32 --geoms 0,1,2 --iters 50 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print('fuse', r['fuse'], 'N', r['N'], 'geometry', r['geometry'], r['us'], 'us')"
done; done
