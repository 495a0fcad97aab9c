#!/bin/bash
# round 4: the remaining BASELINE configs on the final build (config 2: schnell 512^2; config 5 with the host-offload path) and the full-width latent PSNR
# of the GPU path against the oracle-backed twin.   usage: r4_extra.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 900 python tools/latent_parity.py > $O/latent_parity.log 2>&1; tail -3 $O/latent_parity.log; cp gpurun_out/latent_psnr.json $O/ 2>/dev/null
timeout 300 python bench.py --config schnell512 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_schnell512.json 2> $O/bench_schnell512.err
timeout 400 python bench.py --config qwen1024 --offload 2 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_qwen1024_offload2.json 2> $O/bench_qwen1024_offload2.err
timeout 300 python bench.py --deterministic --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024_det.json 2> $O/bench_dev1024_det.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024.json 2> $O/bench_dev1024.err
for f in schnell512 qwen1024_offload2 dev1024_det dev1024; do python3 - $O/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'steps/s', round(d['value'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), 'clock', r.get('effective_clock_ghz'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
