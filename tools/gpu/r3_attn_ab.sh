#!/bin/bash
# same-box A/B of the attention kernel: previous build (tools/ablate/libsvdq_amd_prev.so) vs the tree's.  usage: r3_attn_ab.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2; do
  echo "== prev"; SVDQ_LIB=$PWD/tools/ablate/libsvdq_amd_prev.so PYTHONPATH=$PWD timeout 300 python tools/bench_attention.py 2>&1 | grep -v "^$"
  echo "== new";  PYTHONPATH=$PWD timeout 300 python tools/bench_attention.py 2>&1 | grep -v "^$"
done > $O/attn_ab.txt 2>&1
cat $O/attn_ab.txt
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q > $O/pytest_attn.txt 2>&1; tail -3 $O/pytest_attn.txt
