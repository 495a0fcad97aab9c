#!/bin/bash
# GPU batch 2: compute side alone and memory side alone of the GEMM main loop (cycles per workgroup, effective clock)
O=gpurun_out/batch2; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
$P --lib $AB --shape 4096 12288 3072 --variants 0,11,16,17,18,19,20,21,22,23,24,15,25,26,27,28,29,30 > $O/fc2.jsonl 2> $O/fc2.err
$P --lib $AB --shape 4096 12288 3072 --variants 0,11,15 --no-ws > $O/fc2_nows.jsonl 2>&1
cat $O/fc2.jsonl | cut -c1-20 | head -2
