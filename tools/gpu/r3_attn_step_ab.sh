#!/bin/bash
# same-box A/B of the denoise step with the attention kernel's two geometries.  usage: r3_attn_step_ab.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for g in 1 0; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --attention-geometry $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('attention_geometry', d['config']['attention_geometry'], 'ms/step', round(d['ms_per_step'],2), 'gemm', round(r['gemm_ms_per_step'],2), 'attention', round(r['attention']['ms_per_step'],2), 'TF/s', round(r['attention']['TFLOPs']), 'quant', round(r['quantize']['ms_per_step'],2))"
done; done > $O/attn_step_ab.txt 2>&1
cat $O/attn_step_ab.txt
