#!/bin/bash
O=gpurun_out/trace3; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
$P --lib $AB --shape 4608 3072 9216 --fuse 3 --variants 0 --trace > $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 3072 12288 --fuse 2 --variants 0 --trace >> $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 3072 3072 --variants 0 --trace >> $O/t.jsonl 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/trace3/t.jsonl'):
    r=json.loads(l)
    if 'segments' in r:
        for s in r['segments'][:4]:
            a=[s[0]]+[x for x in s[1:]]
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("   loop %.1f | lora+bias %.1f  fuse-math %.1f  loradown %.1f  stores %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    else: print(f"M={r['M']} K={r['K']} N={r['N']} fuse={r['fuse']}  {r['us']:.1f} us {r['TOPS']:.0f} TOPS")
PY
