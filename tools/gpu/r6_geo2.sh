#!/bin/bash
# round 6: the wave-tile kernel (geometry 8) against the 8-wave kernel (1) after the MFMA changes, isolated launches, rank 32 + bias
for rep in 1 2; do
for s in "4608 3072 3072" "4608 12288 3072" "1536 3072 3072" "1536 12288 3072"; do set -- $s
  timeout 200 tools/ablate/gemm_probe --shape $1 $2 $3 --fuse 0 --R 32 --geoms 1,8 --iters 50 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print('M', r['M'], 'K', r['K'], 'geometry', r['geometry'], r['us'], 'us')"
done; done
