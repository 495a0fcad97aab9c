#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_awq.py tests/test_flux_block_parity.py tests/test_gpu_attention.py -x -q -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert" | tail -15 > gpurun_out/awq_tests.log
timeout 300 python tools/bench_gemv.py > gpurun_out/awq_bench.log 2>&1
cat gpurun_out/awq_tests.log gpurun_out/awq_bench.log
