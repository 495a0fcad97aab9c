#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_norm.py tests/test_flux_block_parity.py tests/test_gpu_attention.py -x -q -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert|rel" | tail -15 > gpurun_out/fused_tests.log; cat gpurun_out/fused_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | cut -c1-330
