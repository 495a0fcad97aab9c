#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_norm.py -q -x 2>&1 | tail -8 > gpurun_out/quant_tests.log
PYTHONPATH=. timeout 300 python tools/bench_quant.py > gpurun_out/quant_bench.log 2>&1
cat gpurun_out/quant_tests.log gpurun_out/quant_bench.log
