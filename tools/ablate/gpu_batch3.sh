#!/bin/bash
# GPU batch 3: one wave per SIMD alone (its partner idles) -- compute-only and memory-only loops without barriers
O=gpurun_out/batch3b; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
# reserved: 3 = no stores / no bias+lowrank epilogue (isolates the loop), 512 = waves 4-7 idle, 1024 = waves 0-3 idle
for r in 3 515 1027; do
$P --lib $AB --shape 4096 12288 3072 --variants 11,16,17,19,27,28,29 --reserved $r >> $O/fc2.jsonl 2>> $O/fc2.err
done
$P --lib $AB --shape 4096 12288 3072 --variants 0,15,30 --reserved 3 >> $O/fc2.jsonl 2>> $O/fc2.err
