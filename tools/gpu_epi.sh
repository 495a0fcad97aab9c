#!/bin/bash
for d in 0 4 1 2 3; do for K in 128 3072; do
  echo "debug=$d K=$K: $(SVDQ_GEMM_DEBUG=$d python tools/bench_kernels.py --iters 10 --shape 4096 $K 9216 2>&1 | grep gemm_us | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print("%.1f us  %.0f TOPS"%(r["gemm_us"], r["gemm_TOPS"]))')"
done; done
