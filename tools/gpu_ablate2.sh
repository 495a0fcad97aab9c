#!/bin/bash
names=(asm cxx no_dma no_lds no_fma no_smfma no_barrier no_dma_lds no_fma_smfma no_dma_barrier compute_only mem_only dma_barrier_only dma_only)
for v in 0 8 10 11 12 13; do
  echo "variant $v ${names[$v]}: $(SVDQ_GEMM_DEBUG=3 SVDQ_GEMM_VARIANT=$v python tools/bench_kernels.py --iters 10 --shape 4096 12288 3072 2>&1 | grep gemm_us | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print("%.1f us  %.0f TOPS"%(r["gemm_us"], r["gemm_TOPS"]))')"
done
