#!/bin/bash
# same-box A/B: eager launches vs the step replayed as one HIP graph
P='import sys,json; r=json.loads(sys.stdin.read()); print("%.2f ms/step" % r["ms_per_step"])'
for rep in 1 2; do
  echo -n "eager: "; timeout 600 python bench.py --steps 20 --warmup 3 --no-prof --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
  echo -n "graph: "; timeout 600 python bench.py --steps 20 --warmup 3 --graph --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
done
