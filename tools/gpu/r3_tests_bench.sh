#!/bin/bash
# all GPU tests + the two bench lines (no profiling).  usage: r3_tests_bench.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -15 $O/pytest_all.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024.json 2> $O/bench_dev1024.err; tail -c 700 $O/bench_dev1024.json
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline > $O/bench_qwen1024.json 2> $O/bench_qwen1024.err; tail -c 500 $O/bench_qwen1024.json
