#!/bin/bash
# One scripted GPU batch (gpurun): parity tests, smoke, micro-benchmarks, kernel timings.
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.txt
tail -6 gpurun_out/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
timeout 300 python tools/debug_fp16.py 2>&1 | tail -16
timeout 600 python tools/bench_kernels.py --iters 20 --json gpurun_out/kernels.json 2>&1 | grep -E '"M": 4096|fused'
