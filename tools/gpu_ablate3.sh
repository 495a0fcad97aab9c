#!/bin/bash
for v in 10 15 16 17 18; do
  echo "variant $v: $(SVDQ_GEMM_DEBUG=3 SVDQ_GEMM_VARIANT=$v python tools/bench_kernels.py --iters 10 --shape 4096 12288 3072 2>&1 | grep gemm_us | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print("%.1f us  %.0f TOPS"%(r["gemm_us"], r["gemm_TOPS"]))')"
done
