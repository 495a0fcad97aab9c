#!/bin/bash
# round 5: GPU test suite (optionally a -k expression / file list), smoke(), the default bench line.  usage: r5_tests_bench.sh <outdir> [pytest args...]
O=gpurun_out/$1; shift; mkdir -p $O
if [ $# -eq 0 ]; then set -- tests; fi
T0=$(date +%s)
timeout 1700 python -m pytest "$@" -m gpu -q --durations=10 > $O/pytest.txt 2>&1; tail -40 $O/pytest.txt; echo "pytest elapsed $(( $(date +%s) - T0 )) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024.json 2> $O/bench_dev1024.err; tail -c 400 $O/bench_dev1024.err
python3 - $O/bench_dev1024.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print('ms/step', round(d['ms_per_step'],2), 'instrumented', d.get('ms_per_step_instrumented'), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), round(r['attention']['frac_bf16'],3), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
