#!/bin/bash
# round 5: Qwen-Image 1024^2 with layer-wise host offload (2 blocks resident), ring depth 2 / 3 / 4 (no depth sweep was recorded).  usage: r5_offload_sweep.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
for S in ${SLOTS:-2 3 4}; do
  timeout 500 python bench.py --config qwen1024 --offload 2 --offload-slots $S --steps 4 --warmup 1 --prof-steps 0 --no-cpu-baseline > $O/bench_qwen1024_offload2_slots$S.json 2> $O/err_$S.txt
  python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_qwen1024_offload2_slots$S.json').read().strip().split('\n')[-1]); print('slots $S: ms/step', round(d['ms_per_step'],1))
except Exception as e: print('slots $S FAILED', e)"
done
