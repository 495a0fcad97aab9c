#!/bin/bash
# timing ablations of the asm main loop (variants >= 2 produce garbage results by design)
names=(asm cxx no_dma no_lds no_fma no_smfma no_barrier no_dma_lds no_fma_smfma no_dma_barrier no_dma_lds_barrier)
for shape in "4096 3072 9216" "4096 12288 3072"; do
for v in 0 1 2 3 4 5 6 7 8 9 10; do
  echo "$shape variant $v ${names[$v]}: $(SVDQ_GEMM_VARIANT=$v python tools/bench_kernels.py --iters 10 --shape $shape 2>&1 | grep gemm_us | python -c 'import sys,json; r=json.loads(sys.stdin.readline()); print("%.1f us  %.0f TOPS"%(r["gemm_us"], r["gemm_TOPS"]))')"
done; done
