#!/bin/bash
# round 4: all GPU tests, smoke(), the default bench line (with cpu_baseline), the config lines (dev / qwen, square and the reference's 1664 x 928 gate,
# 1360 x 768), and the rocprofv3 evidence of the default command.   usage: r4_round.sh <outdir> [skip-tests [no-prof]]
# (the bench lines quote profiles/r4_bench_gemm_*.json with a staleness stamp: after a kernel change run once for the counters, copy them to profiles/, run again with no-prof)
O=gpurun_out/$1; mkdir -p $O
if [ -z "$2" ]; then
  T0=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt; echo "pytest $(( $(date +%s) - T0 )) s"
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024.json 2> $O/bench_dev1024.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --resolution 1360 768 > $O/bench_dev1360x768.json 2> $O/bench_dev1360x768.err
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline > $O/bench_qwen1024.json 2> $O/bench_qwen1024.err
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline --resolution 1664 928 --txt-tokens 37 > $O/bench_qwen1664x928.json 2> $O/bench_qwen1664x928.err
for f in dev1024 dev1360x768 qwen1024 qwen1664x928; do python3 - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), round(r['attention']['frac_bf16'],3), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
if [ -z "$3" ]; then bash tools/gpu/r4_profile_bench.sh $1/prof > $O/prof.log 2>&1; tail -30 $O/prof.log; fi
