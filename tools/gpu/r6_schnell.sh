#!/bin/bash
# round 6: BASELINE config 2 (FLUX.1-schnell 512^2, 1536 tokens): attention geometry 0 (plain grid: 144 workgroups on 256 CUs) vs 2 (persistent: key ranges of the
# 144 tasks dealt over every CU, partial (O, m, l) merged) vs 1; GEMM geometries
O=gpurun_out/$1; mkdir -p $O
P='import sys,json; r=json.loads(sys.stdin.read().strip().split("\n")[-1]); ro=r["roofline"]; print("%.2f ms/step  gemm %.2f (%.3f)  quant %.2f  attn %.2f (%.3f)  %s" % (r["ms_per_step"], ro["gemm_ms_per_step"], ro["frac"], ro["quantize"]["ms_per_step"], ro["attention"]["ms_per_step"], ro["attention"]["frac_bf16"], {k:(round(v["avg_launch_us"],1), round(v["frac"],3)) for k,v in ro["per_variant"].items()}))'
for rep in 1 2; do
  for g in 0 2 1; do echo -n "attention geometry $g: "; timeout 400 python bench.py --config schnell512 --no-cpu-baseline --attention-geometry $g 2>/dev/null | python -c "$P"; done
  for g in 1 2 3; do echo -n "gemm geometry $g: "; timeout 400 python bench.py --config schnell512 --no-cpu-baseline --gemm-geometry $g 2>/dev/null | python -c "$P"; done
done 2>&1 | tee $O/schnell_ab.txt
