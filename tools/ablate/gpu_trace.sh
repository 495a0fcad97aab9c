#!/bin/bash
O=gpurun_out/trace2; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fused_norm.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
$P --lib $AB --shape 4608 3072 9216 --fuse 3 --variants 0 --trace > $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 3072 9216 --variants 0 --trace >> $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 3072 12288 --fuse 2 --variants 0 --trace >> $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 12288 3072 --variants 0 --trace >> $O/t.jsonl 2>&1
$P --lib $AB --shape 4608 3072 3072 --variants 0 --trace >> $O/t.jsonl 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/trace2/t.jsonl'):
    r=json.loads(l)
    if 'segments' in r:
        print("  loop/epi/gap kcyc:", " ".join(f"{(b-a)/1e3:.1f}/{(c-b)/1e3:.1f}" for a,b,c in r['segments']))
    else: print(f"M={r['M']} K={r['K']} N={r['N']} fuse={r['fuse']}  {r['us']:.1f} us {r['TOPS']:.0f} TOPS  {r['eff_GHz']:.2f} GHz")
PY
