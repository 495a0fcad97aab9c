#!/bin/bash
# both attention geometries on one box: timing (tools/bench_attention.py) + the attention tests.  usage: r3_attn64.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
PYTHONPATH=$PWD timeout 300 python tools/bench_attention.py > $O/attn_geometries.txt 2>&1; cat $O/attn_geometries.txt
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $O/pytest_attn.txt 2>&1; tail -15 $O/pytest_attn.txt
