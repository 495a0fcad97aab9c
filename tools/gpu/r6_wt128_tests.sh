#!/bin/bash
# round 6: the product 128 x 64 wave tile kernel: its own tests, the parity suites that now run it by default, same-box step A/B against the round-5 tree.  usage: r6_wt128_tests.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_attention.py -m gpu -q -x -k "wave_tile or fused_output_quantiser" > $O/pytest_wt.txt 2>&1; tail -15 $O/pytest_wt.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_capi.py tests/test_gpu_fused_norm.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; tail -8 $O/pytest_parity.txt
bash tools/gpu_ab_step.sh > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
print('ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
PY
