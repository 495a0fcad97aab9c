#!/bin/bash
# round 5: rank 32 vs rank 128 (and 32 + a rank-16 runtime LoRA) on ONE box: bench lines of the Qwen-Image 1664 x 928 gate shape and of FLUX.1-dev 1024^2.
# usage: r5_rank_ab.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2), r['quantize']['launches'], 'attn', round(r['attention']['ms_per_step'],2), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
run qwen1664x928_r32 --config qwen1024 --resolution 1664 928 --txt-tokens 37
run qwen1664x928_r128 --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128
run qwen1664x928_r32_b --config qwen1024 --resolution 1664 928 --txt-tokens 37
run dev1024_r32 --steps 20 --warmup 3
run dev1024_r32_lora16 --steps 20 --warmup 3 --lora 16
run dev1024_r128 --steps 20 --warmup 3 --rank 128
