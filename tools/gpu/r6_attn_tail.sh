#!/bin/bash
# round 6: the attention kernel's fused quantiser with the low-rank down projection of BOTH row tiles issued before either tile's quantiser arithmetic:
# isolated launch against the library before (gpurun_in/prev), then the parity tests of the attention epilogue
for rep in 1 2 3; do
  for l in gpurun_in/prev/libsvdq_amd.so nunchaku_amd/csrc/libsvdq_amd.so; do
    echo "== $l"; SVDQ_LIB=$PWD/$l RANK0=0 PYTHONPATH=. timeout 200 python tools/bench_attention_tail.py 2>&1 | grep "us"
  done
done
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -x 2>&1 | tail -3
