#!/bin/bash
# round 4: same-box A/B of the weight-code distribution (uniform vs quantised-Gaussian residual) on the default bench; the two-rank bench test.
# usage: r4_codes_ab.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bench_two_ranks.py -m gpu -q > $O/pytest_two_ranks.txt 2>&1; tail -3 $O/pytest_two_ranks.txt
for rep in 1 2; do
  for codes in uniform residual; do
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --weight-codes $codes > $O/bench_${codes}_$rep.json 2> $O/bench_${codes}_$rep.err
    python - <<PY
import json
d = json.loads(open("$O/bench_${codes}_$rep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("$codes $rep ms/step %.2f gemm %.2f frac %.4f attention %.2f clock %s" % (d["ms_per_step"], r["gemm_ms_per_step"], r["frac"], r["attention"]["ms_per_step"], r["effective_clock_ghz"]))
PY
  done
done
