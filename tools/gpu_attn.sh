#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_flux_block_parity.py -q -x 2>&1 | tail -15 > gpurun_out/attn_tests.log
PYTHONPATH=. timeout 300 python tools/bench_attention.py > gpurun_out/attn_bench.log 2>&1
cat gpurun_out/attn_tests.log gpurun_out/attn_bench.log
