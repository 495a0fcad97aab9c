#!/bin/bash
# round 6: deterministic mode, level "runs" (SVDQ_LORA_ACT_Q32_RUNS, ABI 22: fp32 carry inside a row run, fixed point between workgroups): the fc1 launch in the
# three formats; the determinism tests; the bench line at both levels beside the default one
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
  for args in "--q32" "--q32runs" ""; do
    echo -n "fc1 [$args]: "; timeout 120 tools/ablate/gemm_probe --lib nunchaku_amd/csrc/libsvdq_amd.so --shape 4608 3072 12288 --fuse 2 --R 32 --R2 32 $args --geoms 0,1 --iters 50 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print(r['us'], 'us', r['sum'], end=' | ')
print()"
  done
done 2>&1 | tee $O/fc1_ab.txt
timeout 900 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_fused_norm.py tests/test_capi.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_det.txt
for c in "" "--deterministic" "--deterministic runs"; do
  n=$(echo "$c" | tr -d ' -')
  timeout 400 python bench.py --no-cpu-baseline --steps 12 --warmup 2 --prof-steps 5 $c > $O/bench_$n.json 2> $O/bench_$n.err
  python3 - $O/bench_$n.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
PY
done 2>&1 | tee $O/bench_ab.txt
