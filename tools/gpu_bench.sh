#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py --steps 3 --warmup 1 --layers 2 2 --no-cpu-baseline 2>&1 | tail -3
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -3 gpurun_out/bench_r1.err; cat gpurun_out/bench_r1.json
