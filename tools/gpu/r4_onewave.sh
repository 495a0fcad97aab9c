#!/bin/bash
# round 4: what does the second wave of a SIMD buy the GEMM loop?  128 x 128 geometry (4-wave workgroups, 64 x 64 per wave) with two workgroups per CU
# (two waves per SIMD, the product) against ONE workgroup per CU (SVDQ_PROBE_GRID=half: one wave per SIMD), and the 256 x 128 geometry, on a
# shape whose tiles divide evenly (M 4096, N 4096: 1024 / 512 tiles), no stream-K.   usage: r4_onewave.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; L=tools/ablate/libsvdq_amd_probe.so
for K in 3072 12288; do
  for rep in 1 2; do
    echo "== K $K geometry 2 (128x128), two workgroups per CU"; timeout 120 $P --lib $L --shape 4096 $K 4096 --geoms 2 --no-ws --warm 300 --trace
    echo "== K $K geometry 2 (128x128), ONE workgroup per CU"; SVDQ_PROBE_GRID=half timeout 120 $P --lib $L --shape 4096 $K 4096 --geoms 2 --no-ws --warm 300 --trace
    echo "== K $K geometry 1 (256x128)"; timeout 120 $P --lib $L --shape 4096 $K 4096 --geoms 1 --no-ws --warm 300 --trace
  done
done 2>&1 | tee $O/onewave.txt
