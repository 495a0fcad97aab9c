#!/bin/bash
# same-box A/B of v2 loop option variants (ablation library; FUSE_NONE bf16)
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
V=$2
for rep in 1 2; do
$P --lib $AB --shape 4608 12288 3072 --variants $V >> $O/ab.jsonl 2>&1
$P --lib $AB --shape 4608 3072 9216 --variants $V >> $O/ab.jsonl 2>&1
$P --lib $AB --shape 4608 3072 3072 --variants $V >> $O/ab.jsonl 2>&1
done
python3 - $O <<'PY'
import json,sys
names={l.split()[0]:l.split()[1] for l in open('tools/ablate/gen/variants.txt')}
for l in open(sys.argv[1]+'/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(f"M={r['M']} K={r['K']:5d} N={r['N']:5d} v={r['variant']:3d} {names.get(str(r['variant']),'?'):14s} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz")
PY
