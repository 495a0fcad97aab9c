#!/bin/bash
# round 6: all GPU tests, smoke(), the default bench line (with cpu_baseline), the config lines (dev / schnell / qwen, the reference's 1664 x 928 gate at rank 32
# and rank 128, rank 32 + a rank-16 runtime LoRA, deterministic mode) and the rocprofv3 evidence of the default command.   usage: r5_round.sh <outdir> [skip-tests [no-prof]]
O=gpurun_out/$1; mkdir -p $O
if [ -z "$2" ]; then
  T0=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt; echo "pytest $(( $(date +%s) - T0 )) s"
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1200 $O/bench_default.json
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'instr', round(d['ms_per_step_instrumented'] or 0,2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), round(r['attention']['frac_bf16'],3), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
run dev1024 --steps 20 --warmup 3
run dev1024_det --steps 20 --warmup 3 --deterministic
run dev1024_lora16 --steps 20 --warmup 3 --lora 16
run dev1024_r128 --steps 20 --warmup 3 --rank 128
run dev1360x768 --steps 20 --warmup 3 --resolution 1360 768
run schnell512 --config schnell512
run qwen1024 --config qwen1024
run qwen1664x928 --config qwen1024 --resolution 1664 928 --txt-tokens 37
run qwen1664x928_r128 --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128
if [ -z "$3" ]; then bash tools/gpu/r5_profile_bench.sh $1/prof > $O/prof.log 2>&1; tail -30 $O/prof.log; fi
