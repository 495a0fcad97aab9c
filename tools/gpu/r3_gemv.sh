#!/bin/bash
# round 3: the M = 1 path of the AWQ GEMV (loads requested before x is staged; fp16 on the packed pipe): parity, then kernel
# times of both dtypes against the previous build (tools/ablate/libsvdq_amd_prev.so, if present).  usage: tools/gpu/r3_gemv.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_awq.py -m gpu -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
{
for rep in 1 2; do
  if [ -f tools/ablate/libsvdq_amd_prev.so ]; then echo "== previous build"; SVDQ_LIB=tools/ablate/libsvdq_amd_prev.so timeout 300 python tools/bench_gemv.py; fi
  echo "== this build"; timeout 300 python tools/bench_gemv.py
done
} > $O/gemv.txt 2>&1; cat $O/gemv.txt
