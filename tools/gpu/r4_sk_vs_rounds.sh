#!/bin/bash
# round 4: N = 3072 launches (432 tiles of 256 x 128 on 256 CUs): stream-K on 256 workgroups against 216 workgroups x 2 whole tiles (no workspace).   usage: <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; L=nunchaku_amd/csrc/libsvdq_amd.so
for rep in 1 2; do
  for K in 3072 12288; do
    echo "== K $K stream-K (workspace)"; timeout 120 $P --lib $L --shape 4608 $K 3072 --geoms 0 --warm 400
    echo "== K $K whole rounds (no workspace)"; timeout 120 $P --lib $L --shape 4608 $K 3072 --geoms 0 --no-ws --warm 400
  done
done 2>&1 | tee $O/sk_vs_rounds.txt
