#!/bin/bash
# end of round 3: all GPU tests, smoke(), the default bench line (with cpu_baseline) and the Qwen-Image line.  usage: r3_final.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 900 $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_dev1024.json 2> $O/bench_dev1024.err; tail -c 300 $O/bench_dev1024.json
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline > $O/bench_qwen1024.json 2> $O/bench_qwen1024.err; tail -c 300 $O/bench_qwen1024.json
