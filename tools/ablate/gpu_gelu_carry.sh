#!/bin/bash
# GELU_QUANT epilogue: carry of the low-rank-down sums across a workgroup's tiles + column-walk tile order, same-box A/B
# (ablation library; reserved 2048 = no carry (atomics per tile), 4096 = strips enumeration, 6144 = both off = round-2 midpoint)
O=gpurun_out/gelu_carry; mkdir -p $O; rm -f $O/a.jsonl
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
for rep in 1 2; do
for r in 0 2048 4096 6144; do
$P --lib $AB --shape 4608 3072 12288 --fuse 2 --variants 0 --reserved $r >> $O/a.jsonl 2>&1
done; done
$P --lib $AB --shape 4096 3072 12288 --fuse 2 --variants 0 --reserved 0 >> $O/a.jsonl 2>&1
$P --lib $AB --shape 4096 3072 12288 --fuse 2 --variants 0 --reserved 6144 >> $O/a.jsonl 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/gelu_carry/a.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(f"M={r['M']} reserved={r['reserved']:5d}  {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz")
PY
