#!/bin/bash
# round 4: the key mask on attention geometry 2.  parity + the odd-shape bench lines.  usage: r4_attn_mask.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -s -k "mask or padded" > $O/pytest.txt 2>&1; grep -E "max err|passed|failed|Error" $O/pytest.txt | tail -30
timeout 900 python -m pytest tests/test_flux_block_parity.py tests/test_gpu_qwenimage.py -m gpu -q -k "odd or padded" > $O/pytest2.txt 2>&1; tail -3 $O/pytest2.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --resolution 1360 768 > $O/bench_dev1360x768.json 2> $O/bench_dev1360x768.err
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline --resolution 1664 928 --txt-tokens 37 > $O/bench_qwen1664x928.json 2> $O/bench_qwen1664x928.err
for f in dev1360x768 qwen1664x928; do python3 - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), round(r['attention']['frac_bf16'],3))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
