#!/bin/bash
# round 6 ("measure them instead of estimating"): the two QKV-epilogue levers as compile-time timing variants of the probe library (results garbage):
#   rot1 = SVDQ_PROBE_ROT=1: the rotary table read as eight lane-contiguous 16-byte loads per row tile (what a table re-ordered to lane order would give)
#   rot2 = SVDQ_PROBE_ROT=2: no rotary table loads at all (the upper bound of anything that can be done about them)
#   vrow = SVDQ_PROBE_VROW=1: V tiles stored row-major into `out` by the default store path instead of transposed into out_vt
# build: python tools/ablate/build.py;  SVDQ_PROBE_DEFS="-DSVDQ_PROBE_ROT=1" SVDQ_PROBE_TAG=rot1 python tools/ablate/build.py;  ... ROT=2 / rot2;  ... VROW=1 / vrow
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
for rep in 1 2 3 4; do
for v in "" _rot1 _rot2 _vrow; do
  echo -n "QKV 4608x3072x9216 fuse=3 probe${v:-_base}: "
  timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape 4608 3072 9216 --fuse 3 --R 32 --geoms 0,1 --iters 60 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print('geo', r.get('geometry'), r['us'], 'us', round(r.get('eff_GHz',0),3), 'GHz', end=' | ')
print()"
done
done 2>&1 | tee $O/qkv_levers.txt
