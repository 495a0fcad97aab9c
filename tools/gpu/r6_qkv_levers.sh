#!/bin/bash
# round 6 ("measure them instead of estimating"): the two QKV-epilogue levers on the probe library, timing only (SVDQ_PROBE_OFF bits of the
# RMSNORM_ROPE epilogue: 8 = rotary table as lane-contiguous 16-byte loads, 16 = no rotary loads at all, 32 = V tiles row-major instead of transposed into out_vt)
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
for rep in 1 2 3; do
for off in 0 8 16 32 40 48; do
  echo -n "QKV 4608x3072x9216 fuse=3 SVDQ_PROBE_OFF=$off: "
  SVDQ_PROBE_OFF=$off timeout 120 $P --lib $PL --shape 4608 3072 9216 --fuse 3 --R 32 --geoms 0,1 --iters 60 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print('geo', r.get('geometry'), r['us'], 'us', round(r.get('eff_GHz',0),3), 'GHz', end=' | ')
print()"
done
done 2>&1 | tee $O/qkv_levers.txt
