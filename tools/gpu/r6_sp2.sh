#!/bin/bash
# round 6: the 64 x 64 loop with its K-step DMAs spread one unit per tile-group slot ("sp", probe libraries) against the burst, isolated launches, interleaved
for rep in 1 2; do
for s in "4608 3072 3072 0 1" "4608 3072 12288 2 0" "4608 3072 9216 3 0"; do set -- $s
  for l in tools/ablate/libsvdq_amd_probe.so tools/ablate/libsvdq_amd_probe_sp.so; do
    echo -n "fuse=$4 N=$3 $(basename $l): "; timeout 120 tools/ablate/gemm_probe --lib $l --shape $1 $2 $3 --fuse $4 --R 32 --R2 32 --geoms $5 --iters 50 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print(r['us'], 'us', r['wg_cycles'], 'cycles', r['eff_GHz'], 'GHz', r['sum'])"
  done
done; done
