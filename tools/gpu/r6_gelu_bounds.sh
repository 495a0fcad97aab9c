#!/bin/bash
# round 6: bounds of the GELU_QUANT epilogue's low-rank-down block on the probe library (timing only): SVDQ_PROBE_OFF 1 = no carry update / atomics (MFMAs kept),
# 4 = the whole block off, 2 = the code stores off
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
for rep in 1 2 3; do
for off in 0 1 4 2 6; do
  echo -n "fc1 4608x3072x12288 fuse=2 R2=32 SVDQ_PROBE_OFF=$off: "
  SVDQ_PROBE_OFF=$off timeout 120 $P --lib $PL --shape 4608 3072 12288 --fuse 2 --R 32 --R2 32 --geoms 0 --iters 60 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print(r['us'], 'us', round(r.get('eff_GHz',0),3), 'GHz', end=' | ')
print()"
done
done
