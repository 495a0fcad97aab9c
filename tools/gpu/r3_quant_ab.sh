#!/bin/bash
# same-box A/B of the quantiser kernel: previous build (tools/ablate/libsvdq_amd_prev.so) vs the tree's, then its tests.  usage: r3_quant_ab.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
{
for rep in 1 2; do
  echo "== prev"; SVDQ_LIB=$PWD/tools/ablate/libsvdq_amd_prev.so PYTHONPATH=$PWD timeout 200 python tools/bench_quant.py 2>&1 | grep "M="
  echo "== new";  PYTHONPATH=$PWD timeout 200 python tools/bench_quant.py 2>&1 | grep "M="
done
} > $O/quant_ab.txt 2>&1
cat $O/quant_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fused_norm.py tests/test_gpu_geometry_determinism.py -m gpu -q -x 2>&1 | tail -4
