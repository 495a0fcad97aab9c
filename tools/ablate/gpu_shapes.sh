#!/bin/bash
# production shapes of a FLUX step through the product library (+ the quick parity subset)
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PR=nunchaku_amd/csrc/libsvdq_amd.so
if [ "$2" != "notest" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fused_norm.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt; fi
{
for s in "4608 3072 3072" "4608 12288 3072" "512 3072 3072" "512 12288 3072"; do $P --lib $PR --shape $s; done
$P --lib $PR --shape 4608 3072 9216 --fuse 3
$P --lib $PR --shape 4608 3072 12288 --fuse 2
$P --lib $PR --shape 4608 3072 9216 --fuse 3 --split 4096
$P --lib $PR --shape 4608 3072 12288 --fuse 2 --split 4096
$P --lib $PR --shape 4608 3072 3072 --split 4096
$P --lib $PR --shape 4608 12288 3072 --split 4096
} > $O/prod.jsonl 2> $O/prod.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/prod.jsonl'):
    r=json.loads(l); print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} split={r['split']:4d}  {r['us']:7.2f} us  {r['TOPS']:7.1f} TOPS")
PY
