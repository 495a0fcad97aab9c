#!/bin/bash
# per-tile fixed cost vs per-K-step cost: time(K) at fixed M, N
for K in 128 256 512 1024 3072 6144 12288; do
  echo "K=$K: $(python tools/bench_kernels.py --iters 10 --shape 4096 $K ${1:-9216} 2>&1 | grep gemm_us | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print("%.1f us  %.0f TOPS"%(r["gemm_us"], r["gemm_TOPS"]))')"
done
