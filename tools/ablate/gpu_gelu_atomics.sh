#!/bin/bash
# what the low-rank-down atomics of the GELU_QUANT epilogue cost (ablation library: reserved 16 = no atomics, 64 = no low-rank down
# at all, 32 = workgroup-scope atomics; results of those are wrong, timings only)
O=gpurun_out/gelu_atomics; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
for rep in 1 2; do
for r in 0 16 32 64; do
$P --lib $AB --shape 4608 3072 12288 --fuse 2 --variants 0 --reserved $r >> $O/a.jsonl 2>&1
done; done
python3 - <<'PY'
import json
for l in open('gpurun_out/gelu_atomics/a.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(f"reserved={r['reserved']:3d}  {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz")
PY
