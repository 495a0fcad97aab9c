#!/bin/bash
# One scripted GPU batch (gpurun): parity tests, smoke, micro-benchmarks, kernel timings.
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/pytest_gpu.txt
tail -5 gpurun_out/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
timeout 300 tools/ubench > gpurun_out/ubench.jsonl 2>&1; tail -16 gpurun_out/ubench.jsonl
timeout 600 python tools/bench_kernels.py --iters 20 --json gpurun_out/kernels.json 2>&1 | tail -16
