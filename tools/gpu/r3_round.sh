#!/bin/bash
# tests + latent parity + bench + profiles in one call.  usage: tools/gpu/r3_round.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -12 $O/pytest_all.txt
timeout 900 python tools/latent_parity.py > $O/latent_parity.log 2>&1; tail -3 $O/latent_parity.log; cp gpurun_out/latent_psnr.json $O/ 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench_dev1024.json 2> $O/bench_dev1024.err; tail -c 600 $O/bench_dev1024.json
timeout 300 python bench.py --config qwen1024 --no-cpu-baseline > $O/bench_qwen1024.json 2> $O/bench_qwen1024.err; tail -c 400 $O/bench_qwen1024.json
timeout 400 python bench.py --config qwen1024 --offload 2 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_qwen1024_offload2.json 2> $O/bench_qwen1024_offload2.err; tail -c 400 $O/bench_qwen1024_offload2.json
timeout 300 python bench.py --deterministic --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_dev1024_det.json 2> $O/bench_dev1024_det.err; tail -c 300 $O/bench_dev1024_det.json
timeout 900 bash tools/gpu/r3_profile_bench.sh $1/prof > $O/profile.log 2>&1; tail -14 $O/profile.log
timeout 600 bash tools/gpu/r3_attn_step_ab.sh $1 > /dev/null 2>&1; cat $O/attn_step_ab.txt
timeout 300 bash tools/gpu/r3_attn64.sh $1/attn > /dev/null 2>&1; head -8 $O/attn/attn_geometries.txt
