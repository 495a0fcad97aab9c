#!/bin/bash
for mp in 2 4 8; do for ms in 8 4; do
echo "maxpieces=$mp minsteps=$ms"
for shp in "512 12288 3072" "512 3072 3072" "512 3072 9216" "4608 3072 3072"; do
SVDQ_SK_MAXPIECES=$mp SVDQ_SK_MINSTEPS=$ms timeout 120 python tools/bench_kernels.py --shape $shp 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('  ', r['M'], r['K'], r['N'], 'gemm %.1f us' % r['gemm_us'])"
done; done; done
SVDQ_SK_MAXPIECES=8 SVDQ_SK_MINSTEPS=4 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
