#!/bin/bash
# same-box A/B of the denoise step with two builds of the library.  usage: r3_step_ab.sh <outdir> <libA> <libB>
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for lib in $2 $3; do
  SVDQ_LIB=$lib timeout 200 python tools/bench_step_lib.py --steps 10 --warmup 3 --no-cpu-baseline 2>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('$lib', 'ms/step', round(d['ms_per_step'],2), 'gemm', round(r['gemm_ms_per_step'],2), 'frac', round(r['frac'],4), 'attention', round(r['attention']['ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2))"
done; done > $O/step_ab.txt 2>&1
cat $O/step_ab.txt; tail -3 $O/err.txt
