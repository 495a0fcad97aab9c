#!/bin/bash
# round 6 (VERDICT r5 #2 i): the 128 x 64-per-wave / one-wave-per-SIMD loop probe on hardware, validated against the product kernel's output.   usage: r6_gemm128.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm128_probe
{
for s in "512 3072 256" "4608 3072 3072" "4608 12288 3072" "4608 3072 9216" "4608 3072 12288"; do
  set -- $s
  timeout 120 $P --shape $1 $2 $3 --iters 50 || echo "{\"failed\":\"$s rc=$?\"}"
done
timeout 120 $P --shape 4608 3072 3072 --iters 50 --grid 256 || echo "{\"failed\":\"grid256 rc=$?\"}"
timeout 120 $P --shape 4608 3072 3072 --iters 50 --fp16 || echo "{\"failed\":\"fp16 rc=$?\"}"
} > $O/gemm128.jsonl 2> $O/gemm128.err
cat $O/gemm128.jsonl; tail -5 $O/gemm128.err
