#!/bin/bash
# round 6, final build: ONE call on one box -- (1) the rocprofv3 evidence of the default bench command (kernel-trace stats; FETCH_SIZE / WRITE_SIZE / matrix-pipe
# counters in separate --pmc passes) and the probe's phase trace, copied into profiles/ ON THE BOX so that (2) the bench lines behind them quote counters of their own
# kernel sources (bench.py: traffic_stale / stale false), (3) smoke(), (4) every GPU test.  usage: r5_final.sh <outdir> [pytest -k expression instead of the whole suite]   (the caller copies the same files into profiles/)
O=gpurun_out/$1; mkdir -p $O
T0=$(date +%s)
bash tools/gpu/r5_profile_bench.sh $1/prof > $O/prof.log 2>&1; tail -3 $O/prof.log | cut -c1-300
for f in bench_kernel_stats.csv bench_gemm_hbm_counters.json bench_gemm_mfma_util.json; do cp $O/prof/$f profiles/r6_$f; done
echo "prof $(( $(date +%s) - T0 )) s"
bash tools/gpu/r5_gemm_trace.sh $1/tr > $O/trace_summary.txt 2>&1
python tools/epilogue_share.py $O/tr profiles/r6_gemm_epilogue_share.json > $O/epilogue_share.log 2>&1; cp profiles/r6_gemm_epilogue_share.json $O/
python tools/summarize_trace.py $O/tr $O/gemm_phase_trace_final.txt
echo "trace $(( $(date +%s) - T0 )) s"
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'instr', round(d['ms_per_step_instrumented'] or 0,2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), round(r['attention']['frac_bf16'],3), 'clock', r.get('effective_clock_ghz'), 'stale', r.get('traffic_stale'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
run dev1024 --steps 20 --warmup 3
run dev1024_r128 --steps 12 --warmup 2 --prof-steps 5 --rank 128
run dev1024_lora16 --steps 12 --warmup 2 --prof-steps 5 --lora 16
run qwen1664x928 --steps 8 --warmup 2 --prof-steps 4 --config qwen1024 --resolution 1664 928 --txt-tokens 37
run qwen1664x928_r128 --steps 8 --warmup 2 --prof-steps 4 --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128
run dev1024_det --steps 12 --warmup 2 --prof-steps 5 --deterministic
run dev1024_det_runs --steps 12 --warmup 2 --prof-steps 5 --deterministic runs
run schnell512 --config schnell512
run dev1360x768 --steps 12 --warmup 2 --prof-steps 5 --resolution 1360 768
run qwen1024 --steps 8 --warmup 2 --prof-steps 4 --config qwen1024
echo "lines $(( $(date +%s) - T0 )) s"
if [ -z "$2" ]; then timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt
else timeout 600 python -m pytest tests -m gpu -q -k "$2" > $O/pytest_some.txt 2>&1; tail -3 $O/pytest_some.txt; fi
echo "all $(( $(date +%s) - T0 )) s"
