#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -3
for d in ${ATT_VARIANTS:-1024 2048 2049 2050 2056 2064 2072} ; do echo "debug=$d (nw=$((d>>8)) dbg=$((d&255)))"; SVDQ_ATT_DEBUG=$d PYTHONPATH=. timeout 120 python tools/bench_attention.py 2>&1 | grep svdq_att; done
} > gpurun_out/attn_ablate.log 2>&1
cat gpurun_out/attn_ablate.log
